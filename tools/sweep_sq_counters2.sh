#!/bin/bash
# Second SQ counter pass over the one-sweep Arnoldi kernels, per instantiation: average time a vector-memory or an LDS
# instruction is outstanding (SQ_INST_LEVEL_* / instructions), LDS waits and bank conflicts, scalar instructions.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_sq2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pass -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2> $OUT/pass.err
python3 $ROOT/tools/parse_pmc.py $OUT/pass > $OUT/pass.json
python3 - <<PY
import json, re
d = json.load(open("$OUT/pass.json"))
rows = []
for name, c in d.items():
    m = re.search(r"k_arnoldi_(fused|split)<(\d+), *(true|false|\(bool\)[01])", name)
    if not m:
        continue
    g = lambda k: c[k]["mean"]
    wc = g("SQ_WAVE_CYCLES")
    rows.append((int(m.group(2)), m.group(1), g("SQ_INST_LEVEL_VMEM") / max(g("SQ_INSTS_VMEM_RD"), 1.0),
                 g("SQ_INST_LEVEL_LDS") / max(g("SQ_INSTS_LDS"), 1.0), g("SQ_WAIT_INST_LDS") / wc,
                 g("SQ_LDS_BANK_CONFLICT") / wc, g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), wc))
print("# per instantiation; counters are sums over the shader engines")
print("# NV  kernel  cycles a vmem read is outstanding  cycles an LDS instruction is outstanding  wait_inst_lds/wave_cycles  lds_bank_conflict/wave_cycles  salu_insts  lds_insts  vmem_rd_insts  wave_cycles")
for r in sorted(rows):
    print("%3d  %s  %.0f  %.0f  %.3f  %.3f  %.4g  %.4g  %.4g  %.4g" % r)
PY
