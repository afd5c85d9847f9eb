#!/usr/bin/env python3
"""Per kernel family: HBM bytes per launch from the two rocprofv3 counter passes
(FETCH_SIZE, WRITE_SIZE) and the mean duration from the kernel-trace statistics.

usage: make_traffic.py <evidence-dir>   -> JSON on stdout (mesh, m and block size come from bench.json there)

gfx950: FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the kernels that move 16 bytes per lane.
The one-sweep Arnoldi kernels move 8 bytes per lane, a width for which the counters
are not calibrated: their factors come from a micro-benchmark with the same access
pattern and known byte counts (tools/stripbench.hip, mode calib), profiled in the same
evidence run.
"""
import csv
import json
import os
import re
import sys


def family(name):
    if "k_spmv_stencil" in name:
        return "spmv_dots"
    if "k_arnoldi_fused" in name or "k_arnoldi_split" in name:
        return "arnoldi_fused"
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
    # counter factors for 8-byte-per-lane accesses: known bytes / counted KiB
    calib = {}
    strips = bool(cfg.get("basis_strip_major", False))
    cal_file = os.path.join(ev, "calib_pmc.json")
    if os.path.exists(cal_file):
        cpmc = json.load(open(cal_file))
        cs_file = os.path.join(ev, "calib_strips_pmc.json")
        if os.path.exists(cs_file):
            cpmc.update(json.load(open(cs_file)))
        known = {}
        for line in open(os.path.join(ev, "calib_known_bytes.txt")):
            mm = re.match(r"CALIB (strip<[^>]*>) read_bytes (\d+) write_bytes (\d+)", line)
            if mm:
                known[mm.group(1)] = (float(mm.group(2)), float(mm.group(3)))
        for kname, c in cpmc.items():
            for key, (rb, wb) in known.items():
                if key in kname and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    entry = {"fetch_factor": rb / (c["FETCH_SIZE"]["mean"] * 1024.0)}
                    if wb > 0:
                        entry["write_factor"] = wb / (c["WRITE_SIZE"]["mean"] * 1024.0)
                    calib[key] = entry
    f8 = w8 = None
    # the calibration kernels with the basis layout of the run: template argument MODE 4 = strip-major
    mine = {k: e for k, e in calib.items() if k.endswith(", 4>") == strips}
    if mine:
        f8 = sum(e["fetch_factor"] for e in mine.values()) / len(mine)
        # the sweep stores its one result vector with non-temporal 8-byte stores
        w8 = mine.get("strip<16, 1, 256, true, 3, true, %d>" % (4 if strips else 0), {}).get("write_factor")
    for name, c in pmc.items():
        f = fam_of(name)
        if not f or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        n = c["FETCH_SIZE"]["n"]
        if f == "arnoldi_fused":
            if f8 is None or w8 is None:
                continue
            b = (f8 * c["FETCH_SIZE"]["mean"] + w8 * c["WRITE_SIZE"]["mean"]) * 1024.0
        else:
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
        "mesh": mesh, "m": m, "mgs_block": block, "one_sweep_from": cfg.get("one_sweep_from"),
        "generated_bands": bool(cfg.get("generated_bands", False)), "basis_strip_major": strips,
        "calibration_8_byte_lanes": {"per_kernel": calib, "fetch_factor": f8, "write_factor": w8},
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over one benchmark step "
                  "(tools/collect_evidence.sh); per kernel family the dispatch-weighted mean of "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of 16-byte-per-lane reads); "
                  "arnoldi_fused (8 bytes per lane): (fetch_factor*FETCH_SIZE + write_factor*WRITE_SIZE)*1024 with the "
                  "factors of calibration_8_byte_lanes (known bytes / counted KiB of tools/stripbench.hip calib)",
        "hbm_bytes_per_launch": {f: fam_bytes[f] / fam_n[f] for f in fam_bytes},
        "dispatches_in_counter_pass": fam_n,
        "rocprofv3_mean_duration_us": {f: dur[f] / calls[f] / 1e3 for f in dur},
        "rocprofv3_calls": calls,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
