"""ctypes binding of libdiffhe_hip.so (include/diffhe_hip.h).

The library is built in-tree by `__graft_entry__.build()` /
`make -C difffe-physics-lab_amd/csrc`.  There is NO fallback: if the library or a
GPU is missing every solve raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIFFHE_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libdiffhe_hip.so"))

_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_double
ABI_VERSION = 7     # DIFFHE_ABI_VERSION of include/diffhe_hip.h this binding was written against


class MgLevel(C.Structure):
    """struct diffhe_mg_level (include/diffhe_hip.h)."""
    _fields_ = [("nx", _I), ("ny", _I), ("nd", _I), ("reserved", _I), ("vals", _P), ("is_bc", _P), ("vals32", _P),
                ("dense_inv", _P), ("shift", _P), ("rdiag32", _P), ("mask32", _P), ("offdiag16", _P), ("offdiag_scales", _P)]


_LV = C.POINTER(MgLevel)


class AmgLevel(C.Structure):
    """struct diffhe_amg_level (include/diffhe_hip.h)."""
    _fields_ = [("n", _I), ("W", _I), ("vals", _P), ("cols", _P), ("agg", _P), ("agg_ptr", _P), ("agg_members", _P),
                ("vals32", _P), ("agg_weights", _P), ("p_cols", _P), ("p_vals", _P), ("p_width", _I), ("reserved", _I),
                ("dense_inv", _P)]


_AV = C.POINTER(AmgLevel)

# name -> (restype, argtypes); must list every symbol of include/diffhe_hip.h
SIGNATURES = {
    "diffhe_abi_version": (_I, []),
    "diffhe_status_string": (C.c_char_p, [_I]),
    "diffhe_last_hip_error": (C.c_char_p, []),
    "diffhe_traffic_account": (_I, [_I, C.POINTER(_D), C.POINTER(_L)]),
    "diffhe_chain1d_stage_doubles": (_L, [_I, _I, _I, _I]),
    "diffhe_chain1d_solve": (_I, [_P, _P, _L, _L, _P, _L, _P, _I, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "diffhe_chain1d_adjoint": (_I, [_P, _P, _L, _L, _P, _L, _P, _L, _P, _I, _P, _L, _P, _L, _P, _I, _I, _I, _I, _P,
                                    _P]),
    "diffhe_p1_element_integrals": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "diffhe_ell_assemble_rows": (_I, [_P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "diffhe_lattice_assemble_rows": (_I, [_P, _I, _P, _L, _L, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "diffhe_ell_assemble_rows_ref": (_I, [_P, _P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "diffhe_ell_assemble_atomic": (_I, [_P, _P, _I, _P, _L, _L, _P, _P, _I, _I, _I, _I, _P]),
    "diffhe_ell_apply_dirichlet": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "diffhe_ell_spmv_shared": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "diffhe_cg_workspace_doubles": (_L, [_I, _I]),
    "diffhe_ell_cg_solve": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _D, _I, _I, _P, _P, _P, _P, _P]),
    "diffhe_ell_galerkin": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "diffhe_ell_amg_workspace_doubles": (_L, [_AV, _I, _I]),
    "diffhe_ell_amg_pcg_solve": (_I, [_AV, _I, _I, _P, _P, _I, _D, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P]),
    "diffhe_ell_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "diffhe_lattice_pcg_workspace_doubles": (_L, [_LV, _I, _I]),
    "diffhe_lattice_pcg_solve": (_I, [_LV, _I, _I, _P, _P, _P, _I, _D, _D, _I, _I, _I, C.POINTER(_D), _I, _P, _P,
                                      _P, _P, _P, _P, _P]),
    "diffhe_lattice_pcg_profile": (_I, [_I, C.POINTER(_D), C.POINTER(_L)]),
    "diffhe_lattice_kernel_profile": (_I, [_I, C.POINTER(_D), C.POINTER(_L)]),
    "diffhe_lattice_blocks": (_I, [_I, _I]),
    "diffhe_lattice_fused_passes": (_I, []),
    "diffhe_lattice_recompute_ap": (_I, []),
    "diffhe_lattice_apply": (_I, [_LV, _I, _P, _P, _P, _P, _I, _P]),
    "diffhe_lattice_smooth": (_I, [_LV, _I, _P, _P, _P, _P, _D, _I, _P]),
    "diffhe_lattice_cg_step": (_I, [_LV, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "diffhe_lattice_bilinear": (_I, [_LV, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "diffhe_lattice_apply_shared": (_I, [_I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    "diffhe_lattice_restrict_kappa": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "diffhe_lattice_pack_h16": (_I, [_LV, _I, _P, _P, _P, _P, _P]),
    "diffhe_lattice_max_diag": (_I, [_LV, _I, _P, _P]),
    "diffhe_lattice_grad_kappa": (_I, [_I, _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "diffhe_grad_kappa_blocks": (_I, [_I, _I]),
    "diffhe_p1_grad_kappa": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "diffhe_p1_grad_kappa_shared": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "diffhe_to_node_major": (_I, [_P, _L, _P, _P, _I, _I, _I, _P]),
    "diffhe_to_sample_major": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


CHAIN_REFERENCE_ORDER = 1   # DIFFHE_CHAIN_REFERENCE_ORDER


def lib():
    """Load (once) and return the bound library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"libdiffhe_hip.so not found at {LIB_PATH}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.diffhe_abi_version() != ABI_VERSION:
            raise HipExtensionError("libdiffhe_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        L = lib()
        msg = L.diffhe_status_string(status).decode()
        if status == -2:
            msg += ": " + L.diffhe_last_hip_error().decode()
        raise HipExtensionError(f"{what} failed: {msg}")


def ptr(t):
    """Device (or pinned host) address of a tensor, or NULL for None."""
    return None if t is None else C.c_void_p(t.data_ptr())
