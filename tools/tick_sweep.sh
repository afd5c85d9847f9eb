#!/bin/bash
# Per-instantiation sweep durations for several periods of the common store clock (KRY_STORE_TICK, 10 ns ticks).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
for per in "$@"; do
  export KRY_STORE_TICK=$per
  echo "=== KRY_STORE_TICK=$per"
  bash $ROOT/tools/fused_by_nv.sh > $ROOT/gpurun_out/tick_$per.txt 2>&1 || { echo failed; tail -5 $ROOT/gpurun_out/tick_$per.txt; exit 1; }
  cat $ROOT/gpurun_out/fused_nv/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
