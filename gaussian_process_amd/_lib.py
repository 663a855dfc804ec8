"""ctypes binding of libgpmi355x.so (include/gpmi.h).

The library is built in-tree by `gaussian_process_amd/csrc/Makefile`
(`python -m gaussian_process_amd.build` or `__graft_entry__.build()`).  There is
no CPU fallback: if the library is missing or a call fails, the shim raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# The HIP runtime maps streams onto a pool of hardware queues per priority level, 4 by default: the fifth stream of a level
# that gets used shares a queue with an earlier one and their kernels run one after the other (profiles/r04_stream_overlap.txt;
# it cost a second DistGP instance 20 % before its streams were shared).  Eight per level leaves room for three batch lanes,
# a context and the partitioned driver in one process.  Read by the runtime when it initialises: set before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpmi355x.so")

GPMI_OK, GPMI_ERR_NOT_PD, GPMI_ERR_BAD_ARG, GPMI_ERR_RUNTIME = 0, 1, 2, 3
ABI_VERSION = 1

# stage-timer slots (enum in gpmi.h)
T_KBUILD, T_CHOL, T_CHOL_PANEL, T_CHOL_TRAIL, T_LML, T_KS, T_SOLVE_V, T_MEANVAR, \
    T_ALPHA, T_POSTCHOL, T_TRAIL_LAUNCHES, T_TRAIL_FLOPS = range(12)
T_COUNT = 16
TIMER_NAMES = ["kbuild", "chol", "chol_panel", "chol_trail", "lml", "ks", "solve_v", "meanvar",
               "alpha", "postchol", "trail_launches", "trail_flops", "grad"]

_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_vp = C.c_void_p

# name -> (argtypes); every function returns int status unless noted
SIGNATURES = {
    "gpmi_abi_version": [],
    "gpmi_device_count": [C.POINTER(C.c_int)],
    "gpmi_ctx_create": [C.c_int, C.POINTER(_vp)],
    "gpmi_ctx_destroy": [_vp],
    "gpmi_set_option": [_vp, C.c_char_p, _i64],
    "gpmi_rbf": [_vp, _dp, _i64, _dp, _i64, _i64, C.c_double, C.c_double, _dp],
    "gpmi_cov": [_vp, C.c_int, _dp, _i64, _dp, _i64, _i64, C.c_double, C.c_double, _dp],
    "gpmi_set_kernel": [_vp, C.c_int, C.c_double, C.c_double],
    "gpmi_cov_params": [_vp, C.c_int, _dp, _i64, _dp, _i64, _i64, _dp, C.c_int, _dp],
    "gpmi_set_kernel_params": [_vp, C.c_int, _dp, C.c_int],
    "gpmi_set_train": [_vp, _dp, _i64, _i64, _dp],
    "gpmi_factorize": [_vp, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(_i64)],
    "gpmi_fit": [_vp, _dp, _i64, _i64, _dp, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(_i64)],
    "gpmi_get_alpha": [_vp, _dp],
    "gpmi_get_m": [_vp, _dp],
    "gpmi_get_diag": [_vp, _dp],
    "gpmi_get_factor_block": [_vp, _i64, _i64, _i64, _i64, _dp],
    "gpmi_set_test": [_vp, _dp, _i64],
    "gpmi_predict_resident": [_vp, _dp, _dp, C.c_int],
    "gpmi_predict": [_vp, _dp, _i64, _dp, _dp, C.c_int],
    "gpmi_fit_predict_resident": [_vp, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(_i64), _dp, _dp, C.c_int],
    "gpmi_fit_predict_sample_resident": [_vp, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(_i64), _dp, _dp,
                                         C.c_int, _dp],
    "gpmi_post_chol": [_vp, C.c_double, _dp, C.POINTER(_i64)],
    "gpmi_post_sample": [_vp, C.c_double, _dp, _i64, _dp, C.POINTER(_i64)],
    "gpmi_lml_grad": [_vp, _dp, _dp],
    "gpmi_grad_trace": [_vp, _dp, _dp, _i64, _i64, C.c_double, C.c_double, _dp, _dp, _dp, _dp],
    "gpmi_lml_batch": [_vp, _dp, _i64, _dp, C.POINTER(C.c_int)],
    "gpmi_get_timers": [_vp, _dp, C.c_int],
    "gpmi_sync": [_vp],
    "gpmi_probe_mfma_f64": [_vp, _dp],
    "gpmi_probe_mfma_f64_ex": [_vp, C.c_int, C.c_int, C.c_int, _dp],
    "gpmi_probe_gemm": [_vp, _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, _dp],
    "gpmi_probe_resident": [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int],
    "gpmi_probe_hbm_write": [_vp, _i64, _dp],
    "gpmi_probe_hbm_ex": [_vp, _i64, C.c_int, C.c_int, _dp],
    "gpmi_device_info": [_vp, _dp, C.c_int],
    "gpmi_probe_panel": [_vp, C.c_int, _i64, C.c_int, _dp, C.POINTER(C.c_uint64)],
    "gpmi_probe_gemm_beside_server": [_vp, _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _dp],
    "gpmi_probe_stream_overlap": [_vp, C.c_int, C.c_int, C.c_double, _dp],
    "gpmi_probe_launch_storm": [_vp, C.c_int, C.c_int, C.c_double, C.c_int],
    "gpmi_probe_trsv_giveup": [_vp, _i64, C.c_double, C.POINTER(C.c_int), _dp, _dp],
    "gpmi_dev_rbf_rows": [_vp, _vp, _i64, _i64, _i64, _i64, _i64, C.c_double, C.c_double, C.c_double, _vp, _i64],
    "gpmi_dev_rbf_cross": [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, C.c_double, C.c_double, _vp, _i64],
    "gpmi_dev_cov_rows": [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _i64, _i64, _i64, _i64, C.c_double, _vp, _i64],
    "gpmi_dev_cov_cross": [_vp, C.c_int, _dp, C.c_int, _vp, _i64, _vp, _i64, _i64, _i64, C.c_int, _i64, _i64, _vp, _i64],
    "gpmi_dev_potrf_block": [_vp, _vp, _i64, _i64, _i64, _vp],
    "gpmi_dev_trsm_block": [_vp, _vp, _i64, _vp, _i64, _i64, _i64],
    "gpmi_dev_gemm_nt": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, C.c_int, _i64],
    "gpmi_dev_gemm_nt_rowmap": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64],
    "gpmi_dev_gemm_nt_rowmap_host": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _i64],
    "gpmi_dev_gemm_nt_blocks": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _i64],
    "gpmi_dev_logdiag_sumsq": [_vp, _vp, _i64, _i64, _vp, _i64, _vp],
    "gpmi_dev_gemv_t": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "gpmi_dev_trsv_lt": [_vp, _vp, _i64, _vp, _i64],
    "gpmi_dev_trsv_lt_fused": [_vp, _vp, _i64, _vp, _vp, _i64],
    "gpmi_dev_trsv_lt_vinv": [_vp, _vp, _i64, _vp, _vp, _i64, C.c_int],
    "gpmi_dev_trsv_lt_chain": [_vp, _vp, _i64, _vp, _vp, _vp, _i64, C.c_int, _vp],
    "gpmi_dev_set_concurrent": [C.c_int],
    "gpmi_dev_set_option": [C.c_char_p, _i64],
    "gpmi_dev_grad_trace": [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, C.c_double, C.c_double, C.c_double, _vp, _vp],
    "gpmi_dev_row_dots": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "gpmi_comm_load": [C.c_char_p],
    "gpmi_comm_library": [C.c_char_p, _i64, C.POINTER(C.c_int)],
    "gpmi_comm_unique_id": [C.c_char_p],
    "gpmi_comm_create": [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
    "gpmi_comm_destroy": [_vp],
    "gpmi_comm_broadcast": [_vp, _vp, _vp, _i64, C.c_int],
    "gpmi_comm_all_gather": [_vp, _vp, _vp, _vp, _i64],
    "gpmi_comm_all_reduce": [_vp, _vp, _vp, _i64, C.c_int, C.c_int],
    "gpmi_dev_sum_fixed": [_vp, _vp, _i64, _i64, _i64, _vp, C.c_double, _vp],
    "gpmi_dev_axpy2d": [_vp, _vp, _i64, _vp, _i64, _i64, _i64, C.c_double],
}

_lib = None


class GpmiLibraryMissing(RuntimeError):
    pass


def load():
    """Load the shared library once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch (this framework's plumbing for device memory, streams and torch.distributed) ships its own copy of
    # the HIP runtime; whichever copy a process loads first is the one that owns the GPU.  Load torch's first when
    # torch is importable, so the library and torch share one runtime (the other order leaves torch with
    # "No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise GpmiLibraryMissing(
            "libgpmi355x.so not found at %s -- build it with "
            "`python -m gaussian_process_amd.build` (needs hipcc); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the ABI lacks a symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.gpmi_last_error.argtypes = []
    lib.gpmi_last_error.restype = C.c_char_p
    if lib.gpmi_abi_version() != ABI_VERSION:
        raise RuntimeError("libgpmi355x.so ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    return (load().gpmi_last_error() or b"").decode("utf-8", "replace")


def check(status, bad_pivot=None):
    """Map a C status to the exception the reference would raise."""
    if status == GPMI_OK:
        return
    msg = last_error()
    if status == GPMI_ERR_NOT_PD:
        # reference: np.linalg.cholesky raises LinAlgError (GP_regression.py:138,154)
        err = np.linalg.LinAlgError("Matrix is not positive definite")
        err.bad_pivot = bad_pivot
        raise err
    if status == GPMI_ERR_BAD_ARG:
        raise ValueError(msg or "bad argument")
    raise RuntimeError(msg or "HIP runtime error")


def as_f64(a, ndim=None, name="array"):
    arr = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and arr.ndim != ndim:
        raise ValueError("%s must be %d-dimensional, got shape %s" % (name, ndim, arr.shape))
    return arr


def ptr(arr):
    return arr.ctypes.data_as(_dp)


def scalar(x, name="value"):
    """Hyper-parameters may arrive as 0-d / 1-element arrays (l[i] at
    tune_hyperparms_regression.py:369; np.random.uniform(0,5,1) at :408)."""
    a = np.asarray(x, dtype=np.float64)
    if a.size != 1:
        raise ValueError("%s must be a scalar, got shape %s" % (name, a.shape))
    return float(a.reshape(()))
