#!/bin/bash
# Memory-side counters of the one-sweep Arnoldi kernels, per instantiation, in three passes (a pass holds at most
# four TCP_ or four TCC_ counters): (1) vector-cache to L2 read requests, their summed latency and the cycles the
# vector cache stalled with all its miss slots pending or on returning data; (2) L2 to fabric read requests, their
# summed occupancy (level / requests = cycles a request is outstanding), credit stalls towards DRAM and tag
# stalls; (3) address translation in the vector caches: hits, misses and the stalls misses cause.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing"
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o run -- $B "$@" > /dev/null 2> $OUT/p1.err
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o run -- $B "$@" > /dev/null 2> $OUT/p2.err
timeout -s KILL 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_INFLIGHT_MAX GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o run -- $B "$@" > /dev/null 2> $OUT/p3.err
for p in p1 p2 p3; do python3 $ROOT/tools/parse_pmc.py $OUT/$p > $OUT/$p.json; done
python3 - <<PY
import json, re
d = [json.load(open("$OUT/p%d.json" % k)) for k in (1, 2, 3)]
rows = []
for name in d[0]:
    m = re.search(r"k_arnoldi_(fused|split)<(\d+), *(true|false|\(bool\)[01])", name)
    if not m or name not in d[1] or name not in d[2]:
        continue
    g = lambda k, c: d[k][name][c]["mean"] if c in d[k][name] else float("nan")
    act = g(0, "GRBM_GUI_ACTIVE")
    rows.append((int(m.group(2)), m.group(1),
                 g(0, "TCP_TCC_READ_REQ_LATENCY") / max(g(0, "TCP_TCC_READ_REQ"), 1.0),
                 g(0, "TCP_PENDING_STALL_CYCLES") / act, g(0, "TCP_TCR_TCP_STALL_CYCLES") / act,
                 g(1, "TCC_EA0_RDREQ_LEVEL") / max(g(1, "TCC_EA0_RDREQ"), 1.0),
                 g(1, "TCC_EA0_RDREQ_DRAM_CREDIT_STALL") / g(1, "GRBM_GUI_ACTIVE"), g(1, "TCC_TAG_STALL") / g(1, "GRBM_GUI_ACTIVE"),
                 g(2, "TCP_UTCL1_TRANSLATION_MISS") / max(g(2, "TCP_UTCL1_TRANSLATION_HIT") + g(2, "TCP_UTCL1_TRANSLATION_MISS"), 1.0),
                 g(2, "TCP_UTCL1_STALL_MULTI_MISS") / g(2, "GRBM_GUI_ACTIVE"), g(2, "TCP_UTCL1_STALL_INFLIGHT_MAX") / g(2, "GRBM_GUI_ACTIVE"),
                 act))
print("# per instantiation; counters are sums over all instances of a block, GRBM_GUI_ACTIVE = cycles the kernel ran")
print("# NV kernel | cycles per vector-cache read request to L2 | TCP pending stall / active | TCP return-path stall / active |"
      " cycles a fabric read is outstanding | DRAM credit stall / active | L2 tag stall / active |"
      " translation miss share | multi-miss stall / active | in-flight-max stall / active | active cycles")
for r in sorted(rows):
    print("%3d %-5s | %7.0f | %7.2f | %7.2f | %7.0f | %7.3f | %7.3f | %8.5f | %7.3f | %7.3f | %.4g" % r)
PY
