#!/bin/bash
# Approximate-factorisation step: bit-exact tests, wall time per step and rocprofv3 kernel stats at 8192^2.
# usage on the GPU box: bash tools/af_check.sh
set -e
timeout -k 10 300 python -u -m pytest tests/test_approx_fact.py tests/test_pybind_module.py -q -x -m gpu --timeout 200 --timeout-method thread > gpurun_out/af_t.log 2>&1 || { tail -30 gpurun_out/af_t.log; exit 1; }
tail -2 gpurun_out/af_t.log
timeout -k 10 100 python tools/af_time.py 8192
mkdir -p gpurun_out/af
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/af/stats -o run --output-format csv -- python3 $R/tools/af_time.py 8192 > $R/gpurun_out/af/out.txt 2>&1
cd $R
python3 - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/af/stats/run_kernel_stats.csv")))
for r in rows[:6]:
    print("%-50s %5s %12.1f us %6s%%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
