#!/bin/bash
# Per-instantiation durations of the one-sweep Arnoldi kernels under several settings of the
# experimentation knobs (rocprofv3 kernel trace over two steps each).  Usage:
#   tools/sweep_variants.sh name1 "ENV1=.. ENV2=.." name2 "..." ...
# Tables go to gpurun_out/variants/<name>.txt.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/variants
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  OUT=$ROOT/gpurun_out/variants/$name
  rm -rf $OUT; mkdir -p $OUT
  # the variables are exported in a subshell so that rocprofv3 starts python directly
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/bench.json 2> $OUT/err.txt ) || { echo "$name: run failed"; tail -3 $OUT/err.txt; exit 1; }
  python3 - "$OUT" "$name" "$envs" > $ROOT/gpurun_out/variants/$name.txt <<'PY'
import csv, glob, json, re, sys
out, name, envs = sys.argv[1:4]
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows, other = [], []
for r in csv.DictReader(open(f)):
    m = re.search(r"k_arnoldi_(fused|split)<(\d+), *(true|false|\(bool\)[01])", r["Name"])
    if m:
        rows.append((int(m.group(2)), float(r["AverageNs"]) / 1e3, int(r["Calls"]), m.group(3) in ("true", "(bool)1"), m.group(1)))
    else:
        other.append((float(r["TotalDurationNs"]) / 1e6, r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
N = 8.0 * 8194 * 8194
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("# %s  [%s]  ms_per_step %.2f  it/s %.1f" % (name, envs, b["ms_per_step"], b["value"]))
tot = 0.0
for nv, us, calls, gen, kind in sorted(rows):
    by = (1.0 + nv + (2.0 if gen else 5.0) + 1.0) * N
    tot += us
    print("NV=%2d %-5s %s  %8.1f us  %6.1f GB/s  calls %d" % (nv, kind, "gen   " if gen else "stored", us, by / us / 1e3, calls))
print("sum of sweeps per cycle: %.2f ms" % (tot / 1e3))
for t, nm, calls, avg in sorted(other, reverse=True)[:8]:
    print("other: %8.2f ms total  %6d calls  %9.1f us avg  %s" % (t, calls, avg, nm))
PY
  rm -rf $OUT/stats
  head -1 $ROOT/gpurun_out/variants/$name.txt; grep "sum of sweeps" $ROOT/gpurun_out/variants/$name.txt
done
