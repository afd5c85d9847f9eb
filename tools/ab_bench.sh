#!/bin/bash
# A/B of one environment switch on the default bench, alternating twice on the same box:
#   tools/ab_bench.sh NAME=VALUE [bench.py arguments]   -> ms per step and mean sweep duration per run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
cd $ROOT
SW=$1
shift
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $OUT/base_$i.json 2> $OUT/base_$i.err || exit 1
  env $SW timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $OUT/switch_$i.json 2> $OUT/switch_$i.err || exit 1
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("%-14s %8.2f ms/step  roofline frac %s  dominant kernel %s us" % (f.split("/")[-1][:-5], d["ms_per_step"], r.get("frac"), r.get("avg_us")))
PY
