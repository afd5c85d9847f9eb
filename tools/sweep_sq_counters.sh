#!/bin/bash
# SQ counters of the one-sweep Arnoldi kernels, per instantiation (one counter pass over one step).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/pass1 -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2> $OUT/pass1.err
python3 $ROOT/tools/parse_pmc.py $OUT/pass1 > $OUT/pass1.json
python3 - <<PY
import json, re
d = json.load(open("$OUT/pass1.json"))
rows = []
for name, c in d.items():
    m = re.search(r"k_arnoldi_(?:fused|split)<(\d+), *(true|false|\(bool\)[01])", name)
    if not m:
        continue
    g = lambda k: c[k]["mean"]
    wc = g("SQ_WAVE_CYCLES")
    rows.append((int(m.group(1)), g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
                 g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_INSTS_VALU"), wc, g("SQ_BUSY_CYCLES")))
print("# fractions of the wavefronts' lifetime (SQ_WAVE_CYCLES); counters are sums over the shader engines")
print("# NV  wait_any  wait_inst_any  valu_active  lds_active  vmem_active  valu_insts  wave_cycles  busy_cycles")
for r in sorted(rows):
    print("%3d  %.3f  %.3f  %.3f  %.3f  %.3f  %.4g  %.4g  %.4g" % r)
PY
