#!/bin/bash
# Gram-Schmidt block size (links fused per sweep) against mesh size.
set -e
for cfg in "8192 50 3" "2896 50 5" "2048 30 10" "1024 30 10"; do
  set -- $cfg
  for k in 8 16 32; do
    KRY_MGS_BLOCK=$k timeout -k 10 300 python bench.py --mesh $1 --gmres-m $2 --steps $3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/blk_$1_$k.json
    python - <<PY
import json
d=json.load(open("gpurun_out/blk_$1_$k.json"))
print("mesh=$1 m=$2 K=$k it/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]))
PY
  done
done
