#!/bin/bash
# Line solver: k_tdp_apply2 (KRY_TDP_PAIRS=1, default) against k_tdp_apply (KRY_TDP_PAIRS=0) on the 8192^2 S-aniso problem,
# GMRES(6) + line preconditioner.  usage on the GPU box: bash tools/tdp_pairs_check.sh
for w in 1 0; do
  KRY_TDP_PAIRS=$w timeout -k 10 120 python bench.py --problem s_aniso --tdp --gmres-m 6 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/_tdp$w.json 2> gpurun_out/_tdp.err || { tail -5 gpurun_out/_tdp.err; exit 1; }
  python - $w <<'P'
import json,sys
w=sys.argv[1]
d=json.loads(open("gpurun_out/_tdp%s.json"%w).read().strip().splitlines()[-1]); k=d["kernels"]["precond"]
print("KRY_TDP_PAIRS=%s precond avg_ms %s gbps %s step %.2f ms last_max_delta %r" % (w, k["avg_ms"], k["gbps"], d["ms_per_step"], d.get("last_max_delta")))
P
done
