#!/bin/bash
# Round 3: everything measured for profiles/r03*, one call.
#   1. tools/collect_evidence.sh   the default bench line (with cpu_baseline), the same command under rocprofv3
#                                  --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE passes, counter calibrations
#   2. tools/fused_by_nv.sh        per-instantiation durations of the one-sweep kernels
#   3. tools/other_configs.sh      bench lines of the other configurations
#   4. kernel stats of the general-operator (CSR) path and of the line-preconditioned chain under rocprofv3
#   5. memory-side and SQ counters of the sweeps per instantiation
# Usage: round3_evidence.sh [first-part]   (parts 1..5 as above; default 1)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
FROM=${1:-1}
if [ $FROM -le 3 ]; then
bash tools/collect_evidence.sh > gpurun_out/evidence.log 2>&1 || { echo "collect_evidence failed"; tail -5 gpurun_out/evidence.log; exit 1; }
bash tools/fused_by_nv.sh > gpurun_out/fused_by_nv.txt 2> gpurun_out/fused_by_nv.err || { echo "fused_by_nv failed"; exit 1; }
bash tools/other_configs.sh > gpurun_out/other_configs.log 2>&1 || { echo "other_configs failed"; tail -5 gpurun_out/other_configs.log; exit 1; }
fi
mkdir -p gpurun_out/r03x
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03x/csr -o run -- python3 $ROOT/bench.py --operator csr --mesh 8192 --gmres-m 30 --steps 2 --warmup 1 > $ROOT/gpurun_out/r03x/bench_csr_under_rocprof.json 2> $ROOT/gpurun_out/r03x/csr.err || { echo "csr stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03x/cfg5 -o run -- python3 $ROOT/bench.py --problem s_aniso --tdp --fixed-precond --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/r03x/bench_cfg5_under_rocprof.json 2> $ROOT/gpurun_out/r03x/cfg5.err || { echo "cfg5 stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03x/cfg2 -o run -- python3 $ROOT/bench.py --mesh 2048 --gmres-m 30 --steps 100 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/r03x/bench_cfg2_under_rocprof.json 2> $ROOT/gpurun_out/r03x/cfg2.err || { echo "cfg2 stats failed"; exit 1; }
cd $ROOT
bash tools/sweep_mem_counters.sh > gpurun_out/r03x/sweep_mem_table.txt 2> gpurun_out/r03x/sweep_mem.err || echo "mem counters failed"
bash tools/sweep_sq_counters.sh > gpurun_out/r03x/sweep_sq_table.txt 2> gpurun_out/r03x/sweep_sq.err || echo "sq counters failed"
bash tools/sweep_sq_counters2.sh > gpurun_out/r03x/sweep_sq2_table.txt 2> gpurun_out/r03x/sweep_sq2.err || echo "sq counters 2 failed"
find gpurun_out/r03x -name "*kernel_stats.csv" | head
tail -3 gpurun_out/evidence/bench.json | cut -c1-300
