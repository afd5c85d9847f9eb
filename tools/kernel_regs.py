"""Register and scratch use of the kernels in one HIP source, read from the code-object
metadata of a device-only assembly listing:  python tools/kernel_regs.py <file.hip> [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-w",
                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "krylov_amd", "csrc"),
                       "-S", "--cuda-device-only", src, "-o", out])
text = open(out).read()
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    def g(key):
        m = re.search(r"\." + key + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if flt and flt not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%-60s agpr %3s vgpr %3s spill %s scratch %s lds %s" % (dem[:60], blk.split("\n")[0].strip(), g("vgpr_count"),
                                                              g("vgpr_spill_count"), g("private_segment_fixed_size"),
                                                              g("group_segment_fixed_size")))
