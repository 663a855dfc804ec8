"""Host-side handle on one MI355X: NumPy in / NumPy out over the C-ABI.

`GPContext` is the object the drop-in modules (GP_regression.py,
tune_hyperparms_regression.py of this package) funnel through.  It owns a
`gpmi_ctx` (one GPU, its stream and its HBM workspaces); every array crossing
the boundary is a caller-owned float64 C-contiguous NumPy buffer.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr, scalar


class GPContext:
    """One GPU.  Not thread-safe: use one per thread."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.gpmi_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.N = self.d = self.n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpmi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- options / introspection -------------------------------------------------
    def set_option(self, name, value):
        check(self._lib.gpmi_set_option(self._h, name.encode(), int(value)))

    def timers(self):
        buf = np.zeros(_lib.T_COUNT)
        check(self._lib.gpmi_get_timers(self._h, ptr(buf), _lib.T_COUNT))
        return {k: float(buf[i]) for i, k in enumerate(_lib.TIMER_NAMES)}

    def probe_mfma_f64(self):
        v = C.c_double()
        check(self._lib.gpmi_probe_mfma_f64(self._h, C.byref(v)))
        return v.value

    def probe_mfma_f64_ex(self, blocks_per_cu=2, nacc=16, iters=2048):
        """-> (TFLOP/s, shader clock GHz, cycles per MFMA per SIMD)"""
        out = np.zeros(3)
        check(self._lib.gpmi_probe_mfma_f64_ex(self._h, blocks_per_cu, nacc, iters, ptr(out)))
        return tuple(out)

    def probe_gemm(self, M, N, K, lower=0, variant=0, reps=3):
        """-> (TFLOP/s, ms per launch) of the trailing-update GEMM kernel on scratch buffers"""
        out = np.zeros(2)
        check(self._lib.gpmi_probe_gemm(self._h, M, N, K, lower, variant, reps, ptr(out)))
        return tuple(out)

    def device_info(self):
        """what hipDeviceProp_t reports: CUs, clocks (kHz), memory bus width (bits), memory sizes"""
        out = np.zeros(8)
        check(self._lib.gpmi_device_info(self._h, ptr(out), 8))
        keys = ("compute_units", "clock_khz", "mem_clock_khz", "mem_bus_bits", "global_mem_bytes", "l2_bytes",
                "lds_per_workgroup_bytes", "wavefront")
        return {k: float(v) for k, v in zip(keys, out)}

    def probe_panel(self, kind, m=0, reps=20, stamps=False):
        """-> (microseconds per launch, stamps or None): potrf128 (kind 0) / trsm128 on m rows (kind 1) alone"""
        us = C.c_double()
        st = (C.c_uint64 * 64)() if stamps else None
        check(self._lib.gpmi_probe_panel(self._h, int(kind), int(m), int(reps), C.byref(us), st))
        return us.value, (np.array(st, dtype=np.uint64) if stamps else None)

    def probe_trsv_giveup(self, n=1024, wait_ms=200.0):
        """-> (err word, elapsed ms, x): the one-launch backward solve on an identity system whose bottom block is never
        solved -- every wait must run into its bound, set the error word and leave the kernel"""
        err, ms = C.c_int(-1), C.c_double()
        x = np.empty(int(n))
        check(self._lib.gpmi_probe_trsv_giveup(self._h, int(n), float(wait_ms), C.byref(err), C.byref(ms), ptr(x)))
        return err.value, ms.value, x

    def probe_hbm_write(self, nbytes=1 << 30):
        v = C.c_double()
        check(self._lib.gpmi_probe_hbm_write(self._h, int(nbytes), C.byref(v)))
        return v.value

    def probe_hbm_ex(self, nbytes, mode, blocks):
        v = C.c_double()
        check(self._lib.gpmi_probe_hbm_ex(self._h, int(nbytes), int(mode), int(blocks), C.byref(v)))
        return v.value

    # ---- a1: RBF_kernel -------------------------------------------------------------
    def rbf(self, a, b, sigma, l):
        a = as_f64(a, 2, "a")
        b = as_f64(b, 2, "b")
        if a.shape[1] != b.shape[1]:
            raise ValueError("a and b must have the same number of columns (d): %s vs %s"
                             % (a.shape, b.shape))
        out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
        check(self._lib.gpmi_rbf(self._h, ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1],
                                 scalar(sigma, "sigma"), scalar(l, "l"), ptr(out)))
        return out

    KINDS = {"rbf": 0, "lin": 1, "per": 2, "co2": 3}

    def cov(self, kind, a, b, p0, p1=0.0):
        """kernel matrix of the reference's covariance functions: 'rbf' (p0 = sigma, p1 = l),
        'lin' (p0 = c), 'per' (p0 = period, p1 = lengthscale; 1-D inputs)"""
        a = as_f64(a, 2, "a")
        b = as_f64(b, 2, "b")
        if a.shape[1] != b.shape[1]:
            raise ValueError("a and b must have the same number of columns (d): %s vs %s" % (a.shape, b.shape))
        out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
        if kind == "co2":                     # p0 = the 11 hyper-parameters (CO2_example.py:86-89)
            th = as_f64(np.asarray(p0, dtype=np.float64).reshape(-1), 1, "hyperparms")
            check(self._lib.gpmi_cov_params(self._h, 3, ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1],
                                            ptr(th), th.shape[0], ptr(out)))
            return out
        check(self._lib.gpmi_cov(self._h, self.KINDS[kind], ptr(a), a.shape[0], ptr(b), b.shape[0], a.shape[1],
                                 scalar(p0, "p0"), scalar(p1, "p1"), ptr(out)))
        return out

    def set_kernel(self, kind, p0=0.0, p1=0.0):
        """covariance function of the following fit / predict calls (kernel_choice of prediction());
        'co2': p0 = the 11 hyper-parameters of CO2_example.py's covariance_function"""
        if kind == "co2":
            th = as_f64(np.asarray(p0, dtype=np.float64).reshape(-1), 1, "hyperparms")
            check(self._lib.gpmi_set_kernel_params(self._h, 3, ptr(th), th.shape[0]))
            return
        check(self._lib.gpmi_set_kernel(self._h, self.KINDS[kind], scalar(p0, "p0"), scalar(p1, "p1")))

    # ---- fit --------------------------------------------------------------------------
    def set_train(self, X, y):
        X = as_f64(X, 2, "X_train")
        y = as_f64(y, None, "y_train").reshape(-1)
        if y.shape[0] != X.shape[0]:
            raise ValueError("X_train has %d rows but y_train has %d entries" % (X.shape[0], y.shape[0]))
        check(self._lib.gpmi_set_train(self._h, ptr(X), X.shape[0], X.shape[1], ptr(y)))
        self.N, self.d = X.shape

    def factorize(self, sigma, l, noise_var):
        """K + s I -> L, m = L^-1 y; returns the log-marginal-likelihood."""
        lml = C.c_double()
        bad = C.c_int64()
        st = self._lib.gpmi_factorize(self._h, scalar(sigma, "sigma"), scalar(l, "l"),
                                      scalar(noise_var, "noise_var"), C.byref(lml), C.byref(bad))
        check(st, bad.value)
        return lml.value

    def fit(self, X, y, sigma, l, noise_var):
        self.set_train(X, y)
        return self.factorize(sigma, l, noise_var)

    def alpha(self):
        out = np.empty(self.N)
        check(self._lib.gpmi_get_alpha(self._h, ptr(out)))
        return out

    def m(self):
        out = np.empty(self.N)
        check(self._lib.gpmi_get_m(self._h, ptr(out)))
        return out

    def diag(self):
        out = np.empty(self.N)
        check(self._lib.gpmi_get_diag(self._h, ptr(out)))
        return out

    def factor(self, r0=0, r1=None, c0=0, c1=None):
        r1 = self.N if r1 is None else r1
        c1 = self.N if c1 is None else c1
        out = np.empty((r1 - r0, c1 - c0))
        check(self._lib.gpmi_get_factor_block(self._h, r0, r1, c0, c1, ptr(out)))
        return out

    # ---- predict ----------------------------------------------------------------------
    def set_test(self, Xs):
        Xs = as_f64(Xs, 2, "X_test")
        if Xs.shape[1] != self.d:
            raise ValueError("X_test has d=%d but the training set has d=%d" % (Xs.shape[1], self.d))
        check(self._lib.gpmi_set_test(self._h, ptr(Xs), Xs.shape[0]))
        self.n = Xs.shape[0]

    def predict_resident(self, want_sd=True):
        mu = np.empty(self.n)
        o2 = np.empty(self.n)
        check(self._lib.gpmi_predict_resident(self._h, ptr(mu), ptr(o2), 1 if want_sd else 0))
        return mu, o2

    def predict(self, Xs, want_sd=True):
        self.set_test(Xs)
        return self.predict_resident(want_sd)

    def fit_predict_resident(self, sigma, l, noise_var, want_sd=True):
        """prediction() in one pass (GP_regression.py:109-156) for the resident training and test sets: the rows
        K(X*, X) ride through the Cholesky below the y row.  Returns (lml, mu, sd_or_var)."""
        lml = C.c_double()
        bad = C.c_int64()
        mu = np.empty(self.n)
        o2 = np.empty(self.n)
        st = self._lib.gpmi_fit_predict_resident(self._h, scalar(sigma, "sigma"), scalar(l, "l"), scalar(noise_var, "noise_var"),
                                                 C.byref(lml), C.byref(bad), ptr(mu), ptr(o2), 1 if want_sd else 0)
        check(st, bad.value)
        return lml.value, mu, o2

    def fit_predict_sample_resident(self, sigma, l, noise_var, jitter, want_sd=True, want_factor=True):
        """prediction() with its posterior-sample factor in one pass (GP_regression.py:109-156): ONE Cholesky of
        [[K + sI, .], [K(X*, X), K_ss + jitter I]] -- L, v^T and L_ = cholesky(K_ss + jitter I - v^T v) are its blocks.
        Returns (lml, mu, sd_or_var, L_)."""
        lml = C.c_double()
        bad = C.c_int64()
        mu = np.empty(self.n)
        o2 = np.empty(self.n)
        L_ = np.empty((self.n, self.n)) if want_factor else None      # None: it stays on the device (post_sample, post_chol)
        st = self._lib.gpmi_fit_predict_sample_resident(self._h, scalar(sigma, "sigma"), scalar(l, "l"), scalar(noise_var, "noise_var"),
                                                        float(jitter), C.byref(lml), C.byref(bad), ptr(mu), ptr(o2),
                                                        1 if want_sd else 0, ptr(L_) if want_factor else None)
        check(st, bad.value)
        return lml.value, mu, o2, L_

    def fit_predict_sample(self, X, y, Xs, sigma, l, noise_var, jitter, want_sd=True, want_factor=True):
        self.set_train(X, y)
        self.set_test(Xs)
        return self.fit_predict_sample_resident(sigma, l, noise_var, jitter, want_sd, want_factor)

    def fit_predict(self, X, y, Xs, sigma, l, noise_var, want_sd=True):
        self.set_train(X, y)
        self.set_test(Xs)
        return self.fit_predict_resident(sigma, l, noise_var, want_sd)

    def post_chol(self, jitter):
        out = np.empty((self.n, self.n))
        bad = C.c_int64()
        st = self._lib.gpmi_post_chol(self._h, float(jitter), ptr(out), C.byref(bad))
        check(st, bad.value)
        return out

    def post_sample(self, jitter, Z):
        """L_ @ Z with L_ = cholesky(K_ss + jitter I - v^T v) left on the device (GP_regression.py:154-155): Z (n, num_fun)
        are the caller's normals; f_post = mu[:, None] + the result."""
        Z = as_f64(Z, 2, "Z")
        if Z.shape[0] != self.n:
            raise ValueError("Z must have one row per test point: %s for n=%d" % (Z.shape, self.n))
        out = np.empty_like(Z)
        bad = C.c_int64()
        st = self._lib.gpmi_post_sample(self._h, float(jitter), ptr(Z), Z.shape[1], ptr(out), C.byref(bad))
        check(st, bad.value)
        return out

    # ---- f2: LML gradient ------------------------------------------------------------
    def lml_grad(self):
        """(dLML/dl, dLML/dsigma) at the resident factorisation:
        .5*trace((alpha alpha^T - K_y^-1) dK/dtheta) (tune_hyperparms_regression.py:43-57)."""
        dl, ds = C.c_double(), C.c_double()
        check(self._lib.gpmi_lml_grad(self._h, C.byref(dl), C.byref(ds)))
        return dl.value, ds.value

    def grad_trace(self, a, b, sigma, l, alpha, K_y_inv):
        """The same two traces from gradient_ascent's arguments (tune_hyperparms_regression.py:31)."""
        a = as_f64(a, 2, "a")
        b = as_f64(b, 2, "b")
        N, d = a.shape
        if b.shape != (N, d):
            raise ValueError("a and b must both be (N, d): %s vs %s" % (a.shape, b.shape))
        al = as_f64(np.asarray(alpha, dtype=np.float64).reshape(-1), 1, "alpha")
        Ki = as_f64(K_y_inv, 2, "K_y")
        if al.shape[0] != N or Ki.shape != (N, N):
            raise ValueError("alpha must have N entries and K_y must be (N, N)")
        dl, ds = C.c_double(), C.c_double()
        check(self._lib.gpmi_grad_trace(self._h, ptr(a), ptr(b), N, d, scalar(sigma, "sigma"), scalar(l, "l"), ptr(al), ptr(Ki),
                                        C.byref(dl), C.byref(ds)))
        return dl.value, ds.value

    # ---- batched LML ----------------------------------------------------------------
    def lml_batch(self, triples):
        """triples: (T,3) = (l, sigma_f, noise_var) rows.  Returns (lml[T], status[T])."""
        t = as_f64(triples, 2, "triples")
        if t.shape[1] != 3:
            raise ValueError("triples must be (T, 3) = (l, sigma_f, noise_var)")
        out = np.empty(t.shape[0])
        status = np.zeros(t.shape[0], dtype=np.int32)
        check(self._lib.gpmi_lml_batch(self._h, ptr(t), t.shape[0], ptr(out),
                                       status.ctypes.data_as(C.POINTER(C.c_int))))
        return out, status


_tls = threading.local()


def default_context():
    """Per-thread lazily created context on GPMI_DEVICE (default GPU 0)."""
    import os
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or ctx._h is None:
        ctx = GPContext(int(os.environ.get("GPMI_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
        _tls.ctx = ctx
    return ctx
