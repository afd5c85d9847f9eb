#!/bin/bash
# Collects the measured evidence for the default benchmark configuration on one MI355X:
#  1. the bench line itself (with cpu_baseline),
#  2. rocprofv3 --kernel-trace --stats of the same command (without the CPU sample),
#  3. FETCH_SIZE and WRITE_SIZE in two separate counter passes over one step.
# Output goes to gpurun_out/evidence/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/evidence
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.json
# calibration of both counters on 8-byte-per-lane strip accesses with known byte counts
# (the one-sweep Arnoldi kernels use that access width)
timeout -k 10 60 $ROOT/tools/bin/stripbench 8194 calib > $OUT/calib_known_bytes.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o run -- $ROOT/tools/bin/stripbench 8194 calib > /dev/null 2> $OUT/calib_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o run -- $ROOT/tools/bin/stripbench 8194 calib > /dev/null 2> $OUT/calib_write.err
python3 $ROOT/tools/parse_pmc.py $OUT/calib_fetch $OUT/calib_write > $OUT/calib_pmc.json
# ... and on the aligned rows of a strip-major basis (the default layout of a step object's solver on one rank)
timeout -k 10 60 $ROOT/tools/bin/stripbench 8194 calib_strips >> $OUT/calib_known_bytes.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calibs_fetch -o run -- $ROOT/tools/bin/stripbench 8194 calib_strips > /dev/null 2> $OUT/calibs_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calibs_write -o run -- $ROOT/tools/bin/stripbench 8194 calib_strips > /dev/null 2> $OUT/calibs_write.err
python3 $ROOT/tools/parse_pmc.py $OUT/calibs_fetch $OUT/calibs_write > $OUT/calib_strips_pmc.json
find $OUT -name "*.csv" | head -20
python3 $ROOT/tools/make_traffic.py $OUT > $OUT/traffic.json
