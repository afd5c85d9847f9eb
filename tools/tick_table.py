"""Table of the one-sweep kernels' durations by period of the common store clock (gpurun_out/tick_<period>.txt)."""
import re
import sys

def rd(f):
    d = {}
    for l in open(f):
        m = re.match(r"NV=\s*(\d+)\s+\S+ bands\s+([\d.]+) us", l)
        if m:
            d[int(m.group(1))] = float(m.group(2))
    return d

pers = sys.argv[1:]
ds = [rd("gpurun_out/tick_%s.txt" % p) for p in pers]
print("NV", *pers, "best")
for nv in sorted(ds[0]):
    vals = [d.get(nv, 0) for d in ds]
    print(nv, *["%.0f" % v for v in vals], pers[vals.index(min(vals))], "%.3f" % (min(vals) / vals[0]))
print("sum", *["%.0f" % sum(d.values()) for d in ds], "best-per-nv %.0f" % sum(min(d[nv] for d in ds) for nv in ds[0]))
