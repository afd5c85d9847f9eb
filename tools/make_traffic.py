#!/usr/bin/env python3
"""Per kernel family: HBM bytes per launch from the two rocprofv3 counter passes
(FETCH_SIZE, WRITE_SIZE) and the mean duration from the kernel-trace statistics.

usage: make_traffic.py <evidence-dir>   -> JSON on stdout (mesh, m and block size come from bench.json there)

gfx950: FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import json
import os
import re
import sys


def family(name):
    if "k_spmv_stencil" in name:
        return "spmv_dots"
    m = re.search(r"k_mgs_block<(\d+), (\d+), (true|false)", name)
    if m:
        kp, kc, norm = int(m.group(1)), int(m.group(2)), m.group(3) == "true"
        if norm:
            return "mgs_last"
        return None, kp, kc
    return None


def main():
    ev = sys.argv[1]
    cfg = json.load(open(os.path.join(ev, "bench.json")))["config"]
    mesh, m, block = cfg["mesh"][0], cfg["m"], cfg["mgs_block"]
    pmc = json.load(open(os.path.join(ev, "pmc_summary.json")))
    fam_bytes, fam_n = {}, {}
    def fam_of(name):
        f = family(name)
        if isinstance(f, tuple):
            _, kp, kc = f
            return "mgs_block_full" if (kp == block and kc == block) else "mgs_block_part"
        return f
    for name, c in pmc.items():
        f = fam_of(name)
        if not f or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        n = c["FETCH_SIZE"]["n"]
        b = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
        fam_bytes[f] = fam_bytes.get(f, 0.0) + b * n
        fam_n[f] = fam_n.get(f, 0) + n
    dur, calls = {}, {}
    for r in csv.DictReader(open(os.path.join(ev, "stats", "run_kernel_stats.csv"))):
        f = fam_of(r["Name"])
        if f:
            dur[f] = dur.get(f, 0.0) + float(r["TotalDurationNs"])
            calls[f] = calls.get(f, 0) + int(r["Calls"])
    out = {
        "mesh": mesh, "m": m, "mgs_block": block,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over one benchmark step "
                  "(tools/collect_evidence.sh); per kernel family the dispatch-weighted mean of "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of 16-byte-per-lane reads)",
        "hbm_bytes_per_launch": {f: fam_bytes[f] / fam_n[f] for f in fam_bytes},
        "dispatches_in_counter_pass": fam_n,
        "rocprofv3_mean_duration_us": {f: dur[f] / calls[f] / 1e3 for f in dur},
        "rocprofv3_calls": calls,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
