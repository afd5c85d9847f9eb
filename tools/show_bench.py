"""Prints the headline and the per-kernel-class lines of bench.py JSON files."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 1), "ms/step",
          "whole-step HBM frac", d.get("hbm_frac_whole_step"))
    for name, k in d["kernels"].items():
        print("   %-16s launches %4d  avg %8.4f ms  total %8.2f ms  %s GB/s" %
              (name, k["launches"], k["avg_ms"], k["total_ms"], k["gbps"]))
