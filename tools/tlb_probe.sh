#!/bin/bash
# Share of address translations that miss in the compute units' translation caches, per kernel, for a bench.py
# configuration given on the command line (one counter pass over one step).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tlb_probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2> $OUT/p.err
python3 $ROOT/tools/parse_pmc.py $OUT/p > $OUT/p.json
python3 - <<PY
import json
d = json.load(open("$OUT/p.json"))
rows = []
for name, c in d.items():
    g = lambda k: c[k]["mean"] if k in c else float("nan")
    hit, miss = g("TCP_UTCL1_TRANSLATION_HIT"), g("TCP_UTCL1_TRANSLATION_MISS")
    rows.append((g("GRBM_GUI_ACTIVE") * c["GRBM_GUI_ACTIVE"]["n"], name[:90], c["GRBM_GUI_ACTIVE"]["n"], miss / max(hit + miss, 1.0),
                 g("TCP_TCC_READ_REQ_LATENCY") / max(g("TCP_TCC_READ_REQ"), 1.0), g("GRBM_GUI_ACTIVE")))
print("# kernel | dispatches | translation miss share | cycles per vector-cache read request to L2 | active cycles per dispatch")
for r in sorted(rows, reverse=True)[:25]:
    print("%-90s | %4d | %8.5f | %6.0f | %.4g" % r[1:])
PY
