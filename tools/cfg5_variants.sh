#!/bin/bash
# GMRES(50) with the line preconditioner on the anisotropic 8192^2 mesh: the chain's layout and fusion switches
# against each other (two timed steps each) -> gpurun_out/cfg5/.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/cfg5
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 300 python3 bench.py --no-cpu-baseline --problem s_aniso --tdp --fixed-precond --steps 2 --warmup 1"
$B > $OUT/default.json 2> $OUT/default.err || exit 1
KRY_TDP_FUSE_SCALE=0 $B > $OUT/separate_normalisation.json 2> $OUT/b.err || exit 1
KRY_CHAIN_INTERLEAVE=0 $B > $OUT/contiguous_vectors.json 2> $OUT/c.err || exit 1
KRY_CHAIN_INTERLEAVE=0 KRY_TDP_FUSE_SCALE=0 $B > $OUT/contiguous_vectors_separate_normalisation.json 2> $OUT/d.err || exit 1
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d.get("kernels") or {}
    print("%-50s %8.2f ms/step  " % (f.split("/")[-1][:-5], d["ms_per_step"]) +
          "  ".join("%s %.1f" % (n, v["total_ms"] / 2) for n, v in k.items() if v.get("total_ms")))
PY
