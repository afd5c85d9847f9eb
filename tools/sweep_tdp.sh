#!/bin/bash
# Times the line-preconditioner kernels at 8192^2 for several lines-per-wavefront settings.
set -e
for lpw in 16 8 4 2; do
  KRY_TDP_LPW=$lpw timeout -k 10 200 python bench.py --problem s_aniso --tdp --gmres-m 6 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tdp_lpw$lpw.json
  python - <<PY
import json
d=json.load(open("gpurun_out/tdp_lpw$lpw.json"))
k=d["kernels"]["precond"]
print("LPW=$lpw precond avg_ms", k["avg_ms"], "gbps", k["gbps"], "launches", k["launches"])
PY
done
