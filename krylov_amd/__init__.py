"""MI355X-native implementation of the implicit-Euler / GMRES path of
mfdeakin/krylov: HIP kernels behind a C ABI (``include/krylov_amd.h``), a
Python host mirror (``krylov_amd.host``) and the pybind11 module ``krylov``
with the reference's Python surface."""
from . import capi  # noqa: F401

__all__ = ["capi", "host", "problems"]
