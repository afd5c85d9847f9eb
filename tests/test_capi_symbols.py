"""The C-ABI library loads, exports every function include/krylov_amd.h
declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from krylov_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "krylov_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kry_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 40
    for must in ("kry_ctx_create", "kry_op_create_stencil", "kry_op_create_csr", "kry_gmres_solve",
                 "kry_gmres_add_arnoldi", "kry_ie_timestep", "kry_ie_assemble_system"):
        assert must in names


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in declared_functions():
        assert hasattr(L, name), "libkrylov_amd.so does not export %s" % name


def test_python_prototypes_cover_the_header():
    assert sorted(capi.SIGNATURES) == declared_functions()


def test_version_and_error_string():
    L = capi.lib()
    assert b"gfx950" in L.kry_version()
    assert isinstance(L.kry_last_error(), bytes)


def test_no_device_fails_loudly():
    if capi.device_count() > 0:
        pytest.skip("a HIP device is visible")
    h = C.c_void_p()
    st = capi.lib().kry_ctx_create(0, None, C.byref(h))
    assert st != capi.OK and not h.value
    assert b"no HIP device" in capi.lib().kry_last_error()
    with pytest.raises(capi.KrylovError):
        from krylov_amd import host
        host.Context(0)


def test_null_arguments_are_rejected():
    L = capi.lib()
    assert L.kry_ctx_synchronize(None) != capi.OK
    assert L.kry_gmres_solve(None, None, None, None) != capi.OK
    assert L.kry_ie_timestep(None, 0.25, None) != capi.OK
    assert L.kry_op_size(None) == 0


def test_header_is_plain_c(tmp_path):
    """include/krylov_amd.h compiles as C99 with -pedantic (no C++ or HIP types in the boundary)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "krylov_amd.h"\nint main(void) { return kry_version() == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_c_example_links_against_the_library():
    """examples/step_loop.c (plain C caller of the C ABI) builds and resolves its symbols."""
    from krylov_amd import build
    exe = build.build_c_example()
    assert exe and os.path.exists(exe)
