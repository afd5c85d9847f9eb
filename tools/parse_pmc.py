#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name from counter_collection CSVs.

usage: parse_pmc.py <dir-or-csv> [...]   prints JSON {kernel: {counter: mean, n: dispatches}}
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    files = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(a)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name")
                cname = row.get("Counter_Name")
                val = row.get("Counter_Value")
                if not name or not cname or val in (None, ""):
                    continue
                a = acc[name][cname]
                a[0] += float(val)
                a[1] += 1
    out = {k: {c: {"mean": v[0] / v[1], "n": v[1]} for c, v in d.items()} for k, d in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
