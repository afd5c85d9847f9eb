#!/bin/bash
# Copies the summaries of gpurun_out/evidence (tools/collect_evidence.sh) and of
# gpurun_out/fused_by_nv.txt (tools/fused_by_nv.sh) into profiles/ under a prefix:
#   bash tools/publish_evidence.sh r02a_final
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EV=$ROOT/gpurun_out/evidence
P=$ROOT/profiles/$1
cp $EV/bench.json ${P}_bench.json
cp $EV/bench_under_rocprof.json ${P}_bench_under_rocprof.json
cp $EV/stats/run_kernel_stats.csv ${P}_kernel_stats.csv
cp $EV/pmc_summary.json ${P}_pmc_fetch_write_summary.json
cp $EV/calib_pmc.json ${P}_pmc_calibration_8_byte_lanes.json
cp $EV/calib_known_bytes.txt ${P}_calibration_known_bytes.txt
cp $EV/traffic.json $ROOT/profiles/traffic.json
[ -f $ROOT/gpurun_out/fused_by_nv.txt ] && cp $ROOT/gpurun_out/fused_by_nv.txt ${P}_by_instantiation.txt
ls -la ${P}_*
