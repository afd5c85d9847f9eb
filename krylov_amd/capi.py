"""ctypes binding of the C ABI declared in ``include/krylov_amd.h``.

There is no CPU fallback: if ``libkrylov_amd.so`` is missing or no HIP device
is visible, the calls raise.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkrylov_amd.so")

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)

OK = 0
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NO_DEVICE = 1, 2, 3, 4
FIELD_T, FIELD_U, FIELD_V = 0, 1, 2
BOTTOM, TOP, LEFT, RIGHT = 0, 1, 2, 3
GMRES_PLAIN, GMRES_TDP, GMRES_FIXED_PRECOND, GMRES_EXTENDED = 0, 1, 2, 8
K_SPMV, K_MGS, K_MGS_PART, K_MGS_LAST, K_NORMALIZE, K_COMBINE, K_PRECOND, K_ASSEMBLY, K_OTHER, K_FUSED, K_COUNT = range(11)
K_NAMES = ["spmv_dots", "mgs_block_full", "mgs_block_part", "mgs_last", "normalize", "combine", "precond", "assembly",
           "small_solve", "arnoldi_fused"]

# name -> (restype, argtypes); the list mirrors include/krylov_amd.h and is
# checked against the header by the test-suite.
SIGNATURES = {
    "kry_last_error": (C.c_char_p, []),
    "kry_version": (C.c_char_p, []),
    "kry_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kry_ctx_create": (C.c_int, [C.c_int, _vp, _vpp]),
    "kry_ctx_destroy": (None, [_vp]),
    "kry_ctx_synchronize": (C.c_int, [_vp]),
    "kry_ctx_stream": (_vp, [_vp]),
    "kry_ctx_set_comm": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "kry_ctx_memory_info": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kry_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "kry_ctx_set_comm_rccl": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "kry_ctx_comm_is_native": (C.c_int, [_vp]),
    "kry_ctx_comm_selftest": (C.c_int, [_vp]),
    "kry_ctx_rank": (C.c_int, [_vp]),
    "kry_ctx_nranks": (C.c_int, [_vp]),
    "kry_ctx_set_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "kry_ctx_get_kernel_times": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "kry_op_create_stencil": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _vpp]),
    "kry_op_create_csr": (C.c_int, [_vp, C.c_int64, _vp, _vp, _dp, _vpp]),
    "kry_op_create_dense": (C.c_int, [_vp, C.c_int64, _dp, _vpp]),
    "kry_op_destroy": (None, [_vp]),
    "kry_op_size": (C.c_int64, [_vp]),
    "kry_op_apply": (C.c_int, [_vp, _dp, _dp]),
    "kry_op_apply_device": (C.c_int, [_vp, _vp, _vp]),
    "kry_gmres_create": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vpp]),
    "kry_gmres_destroy": (None, [_vp]),
    "kry_gmres_initial_subspace": (C.c_int, [_vp, _dp, _dp]),
    "kry_gmres_assemble_preconditioner": (C.c_int, [_vp, _vp, _dp]),
    "kry_gmres_add_arnoldi": (C.c_int, [_vp, _vp, _dp]),
    "kry_gmres_solve": (C.c_int, [_vp, _vp, _dp, _dp]),
    "kry_gmres_solve_device": (C.c_int, [_vp, _vp, _vp, _vp]),
    "kry_gmres_subspace_vector": (C.c_int, [_vp, C.c_int, _dp]),
    "kry_gmres_hessenberg": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "kry_gmres_hessenberg_matrix": (C.c_int, [_vp, _dp]),
    "kry_gmres_reduced_solve": (C.c_int, [_vp, _dp, C.c_double, _dp]),
    "kry_gmres_prev_result": (C.c_int, [_vp, _dp]),
    "kry_gmres_result_device": (_vp, [_vp]),
    "kry_gmres_apply_preconditioner": (C.c_int, [_vp, _dp, _dp]),
    "kry_gmres_set_mgs_block": (C.c_int, [_vp, C.c_int]),
    "kry_gmres_get_mgs_block": (C.c_int, [_vp]),
    "kry_gmres_set_one_sweep": (C.c_int, [_vp, C.c_int]),
    "kry_gmres_get_one_sweep": (C.c_int, [_vp]),
    "kry_gmres_basis_strip_major": (C.c_int, [_vp]),
    "kry_gmres_basis_chunk_interleaved": (C.c_int, [_vp]),
    "kry_selftest_shared_division": (C.c_int, [_vp, C.c_double, C.c_double, C.c_uint64, C.c_int64, C.c_int,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kry_gmres_set_norm_from_sums": (C.c_int, [_vp, C.c_int]),
    "kry_gmres_get_norm_from_sums": (C.c_int, [_vp]),
    "kry_gmres_last_norm_from_sums": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "kry_gmres_set_tolerance": (C.c_int, [_vp, C.c_double, C.c_int]),
    "kry_gmres_last_stats": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "kry_gmres_size": (C.c_int64, [_vp]),
    "kry_gmres_iterations": (C.c_int, [_vp]),
    "kry_set_one_sweep_split": (C.c_int, [C.c_int]),
    "kry_get_one_sweep_split": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kry_gmres_cycle_bytes": (C.c_int, [_vp, _vp, _dp, C.POINTER(C.c_int)]),
    "kry_ie_create": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.POINTER(C.c_int), C.c_int, C.c_int, _vpp]),
    "kry_ie_create_slab": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                     C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.POINTER(C.c_int), C.c_int, C.c_int, _vpp]),
    "kry_ie_destroy": (None, [_vp]),
    "kry_ie_set_bc_values": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int]),
    "kry_ie_apply_bcs": (C.c_int, [_vp]),
    "kry_ie_init_offdiag": (C.c_int, [_vp]),
    "kry_ie_set_generated_bands": (C.c_int, [_vp, C.c_int]),
    "kry_ie_get_generated_bands": (C.c_int, [_vp]),
    "kry_ie_assemble_system": (C.c_int, [_vp, C.c_double]),
    "kry_ie_timestep": (C.c_int, [_vp, C.c_double, _dp]),
    "kry_ie_apply_delta": (C.c_int, [_vp, _dp, _dp]),
    "kry_apply_delta_noalloc": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "kry_ie_timestep_async": (C.c_int, [_vp, C.c_double]),
    "kry_ie_last_max_delta": (C.c_int, [_vp, _dp]),
    "kry_ie_timestep_approx_fact": (C.c_int, [_vp, C.c_double, _dp]),
    "kry_ie_get_af_system": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, _dp]),
    "kry_ie_use_graph": (C.c_int, [_vp, C.c_int]),
    "kry_ie_time": (C.c_double, [_vp]),
    "kry_ie_num_eqns": (C.c_int64, [_vp]),
    "kry_ie_get_field": (C.c_int, [_vp, C.c_int, _dp]),
    "kry_ie_get_dense_system": (C.c_int, [_vp, _dp]),
    "kry_ie_get_bands": (C.c_int, [_vp, _dp]),
    "kry_ie_get_rhs": (C.c_int, [_vp, _dp]),
    "kry_ie_get_delta": (C.c_int, [_vp, _dp]),
    "kry_ie_local_size": (C.c_int64, [_vp]),
    "kry_ie_get_local": (C.c_int, [_vp, C.c_int, _dp]),
    "kry_ie_solver": (_vp, [_vp]),
    "kry_ie_operator": (_vp, [_vp]),
    "kry_ie_rhs_device": (_vp, [_vp]),
}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int64, C.c_int, _vp)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp)
REDUCE_SUM, REDUCE_MAX = 0, 1
LOCAL_T, LOCAL_U, LOCAL_V, LOCAL_RHS, LOCAL_DELTA = 0, 1, 2, 3, 4

_lib = None


class KrylovError(RuntimeError):
    pass


def lib():
    """Load libkrylov_amd.so and attach the prototypes (raises if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KrylovError(
            "%s is missing: build it with `python -m krylov_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # PyTorch wheels bundle their own HIP runtime.  If this library brought the system
    # runtime into the process first, a later `import torch` would add a second one and
    # PyTorch would see no GPU; importing torch first (when it is installed) makes the
    # two share one runtime.  KRY_NO_TORCH_PRELOAD=1 skips this.
    if "torch" not in sys.modules and not os.environ.get("KRY_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 - absent or broken PyTorch is not this library's concern
            pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(status):
    if status != OK:
        msg = lib().kry_last_error()
        raise KrylovError("krylov_amd status %d: %s" % (status, msg.decode() if msg else ""))


def device_count():
    n = C.c_int(0)
    st = lib().kry_device_count(C.byref(n))
    return n.value if st == OK else 0
