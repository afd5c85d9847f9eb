#!/bin/bash
# Per-instantiation durations of the one-sweep Arnoldi kernels (rocprofv3 kernel trace over two steps).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/fused_nv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"k_arnoldi_(?:fused|roll|split)<(\d+), *(true|false|\(bool\)[01])", r["Name"])
    if m:
        rows.append((int(m.group(1)), float(r["AverageNs"]) / 1e3, int(r["Calls"]), m.group(2) in ("true", "(bool)1")))
N = 8.0 * 8194 * 8194
for nv, us, calls, gen in sorted(rows):
    # the nv + 1 stored vectors and the operator read (two velocity arrays when the
    # interior band entries are evaluated, else five bands), one vector written
    b = (1.0 + nv + (2.0 if gen else 5.0) + 1.0) * N
    print("NV=%2d  %s  %8.1f us  %6.1f GB/s  calls %d" % (nv, "generated bands" if gen else "stored bands   ", us,
                                                        b / us / 1e3, calls))
PY
