#!/bin/bash
# Bench lines of the other configurations (2048^2 GMRES(30) over 100 graph-replayed steps, 16384^2 on one
# GPU, the anisotropic 8192^2 case with the line preconditioner, the restarted solver without and with the line
# preconditioner) -> gpurun_out/configs/.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/configs
mkdir -p $OUT
cd $ROOT
B="timeout -k 10 400 python3 bench.py --no-cpu-baseline"
$B --mesh 2048 --gmres-m 30 --steps 100 --warmup 3 > $OUT/config2_graph_100_steps.json 2> $OUT/config2.err || exit 1
$B --mesh 16384 --steps 2 --warmup 1 > $OUT/config4_one_gpu.json 2> $OUT/config4.err || exit 1
$B --problem s_aniso --tdp --fixed-precond --steps 2 --warmup 1 > $OUT/config5_s_aniso_tdp.json 2> $OUT/config5.err || exit 1
$B --problem s_aniso --tdp --gmres-m 10 --steps 3 --warmup 1 > $OUT/config5_bug_compatible_gmres10.json 2> $OUT/config5b.err || exit 1
$B --extended 1e-8 --steps 2 --warmup 1 > $OUT/extended_s_iso_rtol1e-8.json 2> $OUT/extended.err || exit 1
$B --extended 1e-8 --gmres-m 30 --dt 0.01 --problem s_aniso --tdp --fixed-precond --steps 4 --warmup 1 > $OUT/extended_s_aniso_tdp_gmres30_dt0.01_rtol1e-8.json 2> $OUT/extended_tdp.err || exit 1
for f in $OUT/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('/')[-1], round(d['value'],1), 'it/s', round(d['ms_per_step'],2), 'ms/step', (d.get('roofline') or {}).get('frac'), d['config'].get('basis_strip_major'))
"; done
