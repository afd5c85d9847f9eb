"""In-tree build of the HIP library (gfx950) and of the pybind11 module.

``python -m krylov_amd.build`` compiles
  krylov_amd/libkrylov_amd.so   HIP kernels + the C ABI of include/krylov_amd.h
  krylov_amd/krylov*.so         pybind11 module `krylov` (the reference's Python surface)
hipcc cross-compiles for gfx950 without a GPU being present.
"""
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "krylov_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libkrylov_amd.so")

HIP_SOURCES = ["kry_kernels.hip", "kry_fused.hip", "kry_api.hip"]
# -ffp-contract=off: element-wise arithmetic must round like the reference's
# scalar C++ (no fused multiply-add); reductions call fma() explicitly.
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
             "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build_lib(force=False, verbose=False):
    """Compile every translation unit (in parallel, objects under csrc/_obj) and link the library."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [os.path.join(CSRC, "kry_internal.h"), os.path.join(CSRC, "kry_device.h"),
            os.path.join(ROOT, "include", "krylov_amd.h")]
    if not force and _newer(LIB, srcs + hdrs):
        return LIB
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and _newer(obj, [src] + hdrs):
            continue
        cmd = [hipcc()] + HIP_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def pybind_module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(PKG, "krylov" + suffix)


def build_pybind(force=False, verbose=False):
    src = os.path.join(PKG, "host", "bindpy.cpp")
    if not os.path.exists(src):
        return None
    out = pybind_module_path()
    hdr_dir = os.path.join(PKG, "host")
    deps = [src] + [os.path.join(hdr_dir, f) for f in os.listdir(hdr_dir) if f.endswith(".hpp")]
    deps.append(os.path.join(ROOT, "include", "krylov_amd.h"))
    if not force and _newer(out, deps) and _newer(out, [LIB]):
        return out
    import pybind11
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
           "-I" + os.path.join(ROOT, "include"), "-I" + hdr_dir, src, "-o", out,
           "-L" + PKG, "-lkrylov_amd", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_c_example(force=False, verbose=False):
    """examples/step_loop.c: a plain C99 caller of the C ABI (gcc, no C++)."""
    src = os.path.join(ROOT, "examples", "step_loop.c")
    if not os.path.exists(src):
        return None
    out = os.path.join(ROOT, "examples", "step_loop")
    if not force and _newer(out, [src, os.path.join(ROOT, "include", "krylov_amd.h")]) and _newer(out, [LIB]):
        return out
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           "-L" + PKG, "-lkrylov_amd", "-Wl,-rpath,$ORIGIN/../krylov_amd", "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    build_lib(force, verbose)
    build_pybind(force, verbose)
    build_c_example(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
