"""
ctypes binding of ``libsp_hip.so`` (C ABI: ``include/starry_process_amd.h``).

There is exactly one compute path: the HIP library.  If it is missing or no
MI355X is visible the functions below raise -- nothing falls back to the CPU.
PyTorch is used only for plumbing (device buffers, streams).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsp_hip.so")
# (debug: an A/B variant built by tools/ab_build.sh)
if os.environ.get("SP_LIB_VARIANT"):
    LIB_PATH = os.path.join(_HERE, "libsp_hip_%s.so" % os.environ["SP_LIB_VARIANT"])

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_void_p = ctypes.c_void_p


class sp_star(ctypes.Structure):
    """Mirror of ``sp_star`` in include/starry_process_amd.h."""

    _fields_ = [
        ("period", ctypes.c_double),
        ("inc", ctypes.c_double),
        ("tau", ctypes.c_double),
        ("baseline_var", ctypes.c_double),
        ("baseline_mean", ctypes.c_double),
        ("data_var", ctypes.c_double),
        ("table", ctypes.c_int32),
        ("nobs", ctypes.c_int32),
    ]


STAR_DTYPE = np.dtype(
    [
        ("period", "<f8"),
        ("inc", "<f8"),
        ("tau", "<f8"),
        ("baseline_var", "<f8"),
        ("baseline_mean", "<f8"),
        ("data_var", "<f8"),
        ("table", "<i4"),
        ("nobs", "<i4"),
    ]
)
assert STAR_DTYPE.itemsize == ctypes.sizeof(sp_star) == 56

SP_STAR_NOT_PD, SP_STAR_ZMAX, SP_STAR_NAN, SP_STAR_STALE_PLAN = 1, 2, 4, 8
TEMPORAL = {None: 0, "none": 0, "matern32": 1, "expsquared": 2}

# name -> (restype, argtypes); every symbol the header declares
_D, _I, _L, _V = ctypes.c_double, ctypes.c_int, ctypes.c_long, c_void_p
PROTOTYPES = {
    "sp_version": (_I, []),
    "sp_strerror": (ctypes.c_char_p, [_I]),
    "sp_last_hip_error": (ctypes.c_char_p, []),
    "sp_device_count": (_I, []),
    "sp_create": (_I, [_I, _I, _I, ctypes.POINTER(_V)]),
    "sp_destroy": (None, [_V]),
    "sp_ydeg": (_I, [_V]),
    "sp_udeg": (_I, [_V]),
    "sp_nylm": (_I, [_V]),
    "sp_nwig": (_I, [_V]),
    "sp_stream_synchronize": (_I, [_V, _V]),
    "sp_index_tables": (_I, [_I, _V, _V, _V, _V, _V]),
    "sp_wigner_int_tables": (_I, [_I, _V, _V, _V, _V, _V]),
    "sp_Rx": (_I, [_V, _V, _I, _V, _V, _V]),
    "sp_dotRx": (_I, [_V, _V, _L, _L, _L, _I, _V, _L, _V, _I, _V]),
    "sp_tensordotRz": (_I, [_V, _V, _V, _I, _V, _V]),
    "sp_special_tensordotRz": (_I, [_V, _V, _V, _V, _I, _V, _V]),
    "sp_rTA1": (_I, [_V, _V]),
    "sp_rTA1L": (_I, [_V, _V, _I, _V]),
    "sp_latitude_integrals": (_I, [_I, _D, _D, _V, _V]),
    "sp_gauss_jacobi": (_I, [_I, _D, _D, _V, _V]),
    "sp_gauss_jacobi_grad": (_I, [_I, _D, _D, _V, _V, _V, _V, _V, _V]),
    "sp_allgather_lnlike": (_I, [_V, _V, _V, _I, _V, _V]),
    "sp_tensordotRz_rev": (_I, [_V, _V, _V, _I, _V, _V, _V, _V]),
    "sp_special_tensordotRz_rev": (_I, [_V, _V, _V, _V, _I, _V, _V, _V, _V]),
    "sp_rTA1L_rev": (_I, [_V, _V, _V, _V]),
    "sp_gemm_nt": (_I, [_V, _V, _L, _L, _V, _L, _L, _V, _L, _L, _I, _I, _I, _D, _I, _I, _I, _V]),
    "sp_spd_inverse_workspace_bytes": (ctypes.c_size_t, [_V, _I, _I]),
    "sp_spd_inverse_batched": (_I, [_V, _I, _I, _V, _L, _L, _V, _V, _V, _V, _V]),
    "sp_lnlike_grad_workspace_bytes": (ctypes.c_size_t, [_V, _I, _I, _I]),
    "sp_lnlike_grad_marginal": (_I, [_V, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I, _I, _I, _D, _V, _V, _V, _V, _V, _V]),
    "sp_lnlike_grad_workspace_bytes_multi": (ctypes.c_size_t, [_V, _I, _I, _I, _I]),
    "sp_lnlike_grad_marginal_multi": (_I, [_V, _I, _I, _I, _V, _V, _V, _V, _I, _V, _V, _I, _I, _I, _D, _V, _V, _V, _V,
                                           _V, _V]),
    "sp_gp_condition": (_I, [_V, _I, _I, _V, _V, _V, _V, _V, _V, _V]),
    "sp_alpha_beta": (_I, [_D, _I, c_double_p, c_double_p, c_double_p, c_double_p]),
    "sp_set_marginal_constants": (_I, [_V, _V, _V]),
    "sp_set_ylm_moments": (_I, [_V, _V, _V]),
    "sp_set_ylm_moments_dev": (_I, [_V, _V, _V, _V]),
    "sp_get_polar_moments": (_I, [_V, _V, _V]),
    "sp_ylm_moments_quadrature": (_I, [_V, _V, _I, _I, _V, _V, _I, _I, _D, _D, _D, _D,
                                       _V, _V, _V]),
    "sp_ylm_moments_quadrature_grad": (_I, [_V, _V, _V, _V, _V, _V, _V, _I, _I, _D, _D, _D, _D,
                                            _V, _V, _V, _V, _V]),
    "sp_debug_panel2_trace": (_I, [_V]),
    "sp_debug_panel2_chain": (_I, [_V]),
    "sp_debug_asm_chunks": (_I, [_I, _I, _V]),
    "sp_debug_set_syrk128_from": (_I, [_I]),
    "sp_debug_set_small_k": (_I, [_I]),
    "sp_debug_set_syrk_symdiag": (_I, [_I]),
    "sp_debug_set_look_ahead": (_I, [_V, _I]),
    "sp_debug_set_panel_layout": (_I, [_V, _I, _I]),
    "sp_profile_kind": (_I, [_V, _I, ctypes.POINTER(ctypes.c_long), c_double_p, c_double_p]),
    "sp_profile_kind_ex": (_I, [_V, _I, ctypes.POINTER(ctypes.c_long), c_double_p, c_double_p, c_double_p]),
    "sp_set_lazy_cov": (_I, [_V, _I]),
    "sp_set_defer_norm": (_I, [_V, _I]),
    "sp_profile_begin_kinds": (_I, [_V, _I, ctypes.c_uint]),
    "sp_profile_begin": (_I, [_V, _I]),
    "sp_profile_end": (_I, [_V, ctypes.POINTER(ctypes.c_long), c_double_p, c_double_p]),
    "sp_kernel_table": (_I, [_V, _V, _I, _I, _V, _V, _V, _V]),
    "sp_cov_marginal_batched": (
        _I, [_V, _I, _I, _V, _V, _I, _V, _V, _I, _I, _I, _V, _L, _L, _V, _V]),
    "sp_design_matrix": (_I, [_V, _I, _I, _V, _V, _V, _V, _V]),
    "sp_cov_conditional_batched": (
        _I, [_V, _I, _I, _V, _V, _V, _I, _I, _I, _V, _L, _L, _V, _V, _V]),
    "sp_cho_factor": (_I, [_V, _V, _I, _L, _L, _I, _V, _V]),
    "sp_cho_solve": (_I, [_V, _V, _I, _L, _L, _V, _I, _I, _V]),
    "sp_tri_solve": (_I, [_V, _V, _I, _L, _L, _V, _I, _I, _I, _V]),
    "sp_solve_rev": (_I, [_V, _V, _I, _L, _L, _V, _V, _I, _I, _I, _V, _V, _V]),
    "sp_cholesky_rev": (_I, [_V, _V, _I, _L, _L, _V, _I, _V, _V]),
    "sp_lnlike_workspace_bytes": (_L, [_V, _I, _I, _I]),
    "sp_lnlike_ensemble": (
        _I, [_V, _I, _I, _I, _V, _V, _V, _V, _I, _I, _V, _V, _V, _I, _I, _I, _D,
             _V, _V, _V, _V]),
    "sp_cholesky_lnlike_batched": (_I, [_V, _I, _I, _I, _V, _V, _V, _V, _V, _V]),
    "sp_plan_data": (_I, [_V, _I, _I, _I, _V, _V, _V, _V, _I, _I, _V, _V, ctypes.POINTER(_V)]),
    "sp_plan_destroy": (None, [_V]),
    "sp_plan_replicate": (_I, [_V, _V, _I, _V, ctypes.POINTER(_V)]),
    "sp_plan_systems": (_I, [_V]),
    "sp_set_size_basis": (_I, [_V, _V, _V, _I, _D]),
    "sp_polar_moments_samples": (_I, [_V, _I, _V, _D, _D, _V, _V, _V]),
    "sp_kernel_table_samples": (_I, [_V, _I, _V, _V, _V, _I, _I, _V, _V, _V, _V]),
    "sp_plan_get_wbar": (_I, [_V, _V]),
    "sp_lnlike_ensemble_planned": (_I, [_V, _V, _V, _V, _V, _V, _V, _V, _I, _D, _V, _V, _V, _V]),
}

_lib = None


class SPError(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SPError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C starry_process_amd/csrc` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH
            )
        # PyTorch bundles its own HIP runtime (same soname, libamdhip64.so.7).
        # It must be the first one loaded so that this library binds to the
        # runtime that owns the tensors' device memory and streams; loading
        # /opt/rocm's copy first would put two runtimes in one process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        L = lib()
        msg = L.sp_strerror(status).decode()
        if status == -2:
            msg += " (" + L.sp_last_hip_error().decode() + ")"
        raise SPError("libsp_hip: %s" % msg)


def hptr(a):
    """Pointer to a C-contiguous NumPy array (host)."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_void_p)


def index_tables(ydeg):
    """Integer layout tables (host only, no GPU needed)."""
    N = (ydeg + 1) ** 2
    out = dict(
        l_of=np.empty(N, np.int32),
        m_of=np.empty(N, np.int32),
        mirror=np.empty(N, np.int32),
        m0=np.empty(ydeg + 1, np.int32),
        blk=np.empty(ydeg + 2, np.int32),
    )
    check(lib().sp_index_tables(ydeg, *[hptr(out[k]) for k in ("l_of", "m_of", "mirror", "m0", "blk")]))
    return out


def wigner_int_tables(ydeg):
    names = ("cosmal", "sinmal", "sgn", "cosmga", "sinmga")
    out = {k: np.empty(ydeg + 1, np.int32) for k in names}
    check(lib().sp_wigner_int_tables(ydeg, *[hptr(out[k]) for k in names]))
    return out


def alpha_beta(z, order=20):
    v = [ctypes.c_double() for _ in range(4)]
    check(lib().sp_alpha_beta(float(z), int(order), *[ctypes.byref(x) for x in v]))
    return tuple(x.value for x in v)
