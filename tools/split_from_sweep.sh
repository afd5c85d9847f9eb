#!/bin/bash
# Per-instantiation sweep durations for several first steps of the 512-thread kernel (KRY_SPLIT_FROM).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  export KRY_SPLIT_FROM=$v
  bash $ROOT/tools/fused_by_nv.sh > $ROOT/gpurun_out/tick_split$v.txt 2>&1 || { echo failed; exit 1; }
  tail -1 $ROOT/gpurun_out/fused_nv/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
