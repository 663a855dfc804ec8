"""CPU oracle for the GP-regression hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This module is a NumPy restatement of the reference's algorithm for the path
named in BASELINE.json (`north_star`): squared-exponential kernel matrix,
K + s*I, Cholesky, the two solves for alpha, predictive mean / variance and the
log-marginal-likelihood.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product package
`gaussian_process_amd` never does (it fails loudly when the HIP library is
missing instead of falling back to this file).

Every function cites the reference lines it follows (paths are relative to
/root/reference, which does not exist on the GPU box; nothing here reads it).

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference ITSELF, generated in the
build container by `oracle/make_golden.py` (which imports
/root/reference/GP_regression.py) and committed under `tests/golden/`.
`tests/test_oracle_vs_golden.py` checks this file against those vectors.
`tune_hyperparms_regression.py` and `CO2_example.py` are Python-2 modules that cannot be
imported whole; make_golden.py executes the source text of their individual functions
(`compute_mar_likelihood`, `gradient_ascent`, `bayesian_opt`, `UCB`, `EI`, `TS`, `overlap`,
and the CO2 example's `covariance_function`, `compute_mar_likelihood`, `bayesian_opt`,
`make_prediction`), which are valid Python 3 on their own, and -- through lib2to3's print / repr
fixers -- the ones with Python-2 print statements (`PI`, `random_gen_test_parms`,
`tune_hyperparms_second`, the CO2 example's `tune_hyperparameters_BO`); the restatements here and
the drop-ins' host logic are checked against those outputs (tests/golden/kernels_*.npz).
`tune_hyperparms_first` is restated and pinned through the functions it calls.
"""
from __future__ import annotations

import numpy as np

# Constants hard-coded inside the reference's function bodies.
NOISE_VAR = 0.0005        # GP_regression.py:120, tune_hyperparms_regression.py:302
SIGMA_F = 1               # GP_regression.py:121
POST_JITTER = 1e-6        # GP_regression.py:154
BO_NOISE_VAR = 0.0001     # tune_hyperparms_regression.py:75


def RBF_kernel(a, b, sigma, l):
    """GP_regression.py:8-19 -- broadcast (N,d,M) difference, square, sum over
    axis 1 (sequential in k), then sigma**2 * exp(-.5 * (1/l**2) * sqdist)."""
    sqdist = ((a[:, :, None] - b[:, :, None].T) ** 2).sum(1)
    return (sigma ** 2) * np.exp(-.5 * (1 / (l ** 2)) * sqdist)


def RBF_kernel_chunked(a, b, sigma, l, rows=256):
    """Same per-element arithmetic as GP_regression.py:18-19 but row-chunked so
    the (N,d,M) temporary stays small (memory-feasible variant)."""
    a = np.asarray(a)
    b = np.asarray(b)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
    bt = b[:, :, None].T
    for r0 in range(0, a.shape[0], rows):
        blk = a[r0:r0 + rows]
        sq = ((blk[:, :, None] - bt) ** 2).sum(1)
        out[r0:r0 + rows] = (sigma ** 2) * np.exp(-.5 * (1 / (l ** 2)) * sq)
    return out


def lin_kernel(a, b, c):
    """GP_regression.py:22-33"""
    output_variance = 1
    fun_mean = 0
    dot_product = np.dot(a - c, b.T - c)
    return fun_mean + output_variance * dot_product


def per_kernel(a, b, parameters):
    """GP_regression.py:36-50 (1-D inputs: np.tile at :48)"""
    output_variance = 1
    p, l = parameters
    num_a = len(a)
    num_b = len(b)
    l2_norm = np.absolute(np.tile(a, (1, num_b)) - np.tile(b.T, (num_a, 1)))
    return output_variance * np.exp(-2 * (np.sin(np.pi * l2_norm / p)) ** 2 / l ** 2)


def prediction_other(X_train, X_test, y_train, kernel_choice, l, num_fun):
    """GP_regression.py:109-156, 'lin' / 'per' branches (:129-136)"""
    kern = lin_kernel if kernel_choice == 'lin' else per_kernel
    s = NOISE_VAR
    N, n = len(X_train), len(X_test)
    K_train, K_s, K_ss = kern(X_train, X_train, l), kern(X_train, X_test, l), kern(X_test, X_test, l)
    L = np.linalg.cholesky(K_train + s * np.eye(N))
    m = np.linalg.solve(L, y_train)
    alpha = np.linalg.solve(L.T, m)
    mu_post = np.dot(K_s.T, alpha)
    v = np.linalg.solve(L, K_s)
    var_test = np.diag(K_ss) - np.sum(v ** 2, axis=0)
    with np.errstate(invalid='ignore'):
        stand_devi = np.sqrt(var_test)
    L_ = np.linalg.cholesky(K_ss + POST_JITTER * np.eye(n) - np.dot(v.T, v))
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(n, num_fun)))
    return mu_post, stand_devi, f_post_fun


_C_LIB = None


def _c_lib():
    """oracle/build/librbf_oracle.so (rbf_oracle.c), built by oracle/Makefile."""
    global _C_LIB
    if _C_LIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "librbf_oracle.so")
        lib = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        lib.rbf_oracle.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                   C.c_double, dp, C.c_int64]
        lib.rbf_oracle.restype = None
        _C_LIB = lib
    return _C_LIB


def RBF_kernel_c(a, b, sigma, l, diag_add=0.0):
    """GP_regression.py:18-19 through rbf_oracle.c (same per-element arithmetic,
    multi-threaded, no (N,d,M) temporary); diag_add folds the `+ s*np.eye(N)` of :138."""
    import ctypes as C
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty((a.shape[0], b.shape[0]))
    dp = C.POINTER(C.c_double)
    coef = -.5 * (1 / (float(l) ** 2))
    _c_lib().rbf_oracle(a.ctypes.data_as(dp), a.shape[0], b.ctypes.data_as(dp), b.shape[0], a.shape[1],
                        coef, float(sigma) ** 2, float(diag_add), out.ctypes.data_as(dp), b.shape[0])
    return out


def dataset_generator(N, n):
    """GP_regression.py:53-68 -- consumes np.random's global state in the order
    uniform(N,1) then randn(N)."""
    s = 0.0005
    f = lambda x: np.sin(0.9 * x).flatten()
    X_train = np.random.uniform(-5, 5, size=(N, 1))
    y_train = f(X_train) + np.sqrt(s) * np.random.randn(N)
    X_test = np.linspace(-5, 5, n).reshape(-1, 1)
    return f, X_train, y_train, X_test


def posterior(X_train, X_test, y_train, sigma, l, s):
    """GP_regression.py:126-148 with free (sigma, s): returns every intermediate
    the parity tests compare (L, m, alpha, mu, v, var)."""
    N = len(X_train)
    K_train = RBF_kernel(X_train, X_train, sigma, l)             # :126
    K_s = RBF_kernel(X_train, X_test, sigma, l)                  # :127
    K_ss = RBF_kernel(X_test, X_test, sigma, l)                  # :128
    L = np.linalg.cholesky(K_train + s * np.eye(N))              # :138
    m = np.linalg.solve(L, y_train)                              # :139
    alpha = np.linalg.solve(L.T, m)                              # :140
    mu_post = np.dot(K_s.T, alpha)                               # :143
    v = np.linalg.solve(L, K_s)                                  # :144
    var_test = np.diag(K_ss) - np.sum(v ** 2, axis=0)            # :147
    return dict(K=K_train, K_s=K_s, K_ss=K_ss, L=L, m=m, alpha=alpha,
                mu=mu_post, v=v, var=var_test)


def prediction(X_train, X_test, y_train, kernel_choice, l, num_fun):
    """GP_regression.py:109-156 (rbf branch only; 'lin'/'per' are out of scope,
    SURVEY.md section 2 row 4).  Draws np.random.normal((n, num_fun)) once (:155)."""
    if kernel_choice != 'rbf':
        raise NotImplementedError("oracle restates the 'rbf' branch only")
    s = NOISE_VAR
    sigma = SIGMA_F
    n = len(X_test)
    p = posterior(X_train, X_test, y_train, sigma, l, s)
    with np.errstate(invalid='ignore'):
        stand_devi = np.sqrt(p['var'])                           # :148
    L_ = np.linalg.cholesky(p['K_ss'] + POST_JITTER * np.eye(n)
                            - np.dot(p['v'].T, p['v']))          # :154
    f_post_fun = p['mu'].reshape(-1, 1) + np.dot(
        L_, np.random.normal(size=(n, num_fun)))                 # :155
    return p['mu'], stand_devi, f_post_fun


def compute_mar_likelihood(X_train, X_test, y_train, sigma, l, s=NOISE_VAR):
    """tune_hyperparms_regression.py:292-313.  X_test is accepted and unused, as
    in the reference.  `s` defaults to the hard-coded 0.0005 (:302)."""
    n = len(X_train)                                             # :303
    K_train = RBF_kernel(X_train, X_train, sigma, l)             # :306
    L = np.linalg.cholesky(K_train + s * np.eye(n))              # :307
    m = np.linalg.solve(L, y_train)                              # :308
    alpha = np.linalg.solve(L.T, m)                              # :309
    return (-.5 * np.dot(y_train.T, alpha)
            - np.log(np.diagonal(L)).sum(0)
            - n / 2.0 * np.log(2 * np.pi))                       # :312


def bayesian_opt(X_train, X_test, y_train):
    """tune_hyperparms_regression.py:67-101: the same posterior with s=1e-4,
    sigma=l=1, num_fun=1."""
    s = BO_NOISE_VAR
    n = len(X_test)
    p = posterior(X_train, X_test, y_train, 1, 1, s)
    with np.errstate(invalid='ignore'):
        stand_devi = np.sqrt(p['var'])                           # :95
    L_ = np.linalg.cholesky(p['K_ss'] + 1e-6 * np.eye(n)
                            - np.dot(p['v'].T, p['v']))          # :98
    f_post_fun = p['mu'].reshape(-1, 1) + np.dot(
        L_, np.random.normal(size=(n, 1)))                       # :99
    return p['mu'], stand_devi, f_post_fun


# ---------------------------------------------------------------------------
# gradient-ascent tuner
# ---------------------------------------------------------------------------
GA_STEP_SIZE = 0.01          # tune_hyperparms_regression.py:42
GA_TOLERANCE = 0.001         # tune_hyperparms_regression.py:117
GA_MAX_ITER = 10000          # tune_hyperparms_regression.py:121


def lml_gradient_terms(a, b, sigma, l, alpha, K_y):
    """tune_hyperparms_regression.py:43-57: (l_var, sigma_var).  l_var is the live code
    (:54-57); sigma_var is the commented-out twin (:46-51), restated with the same calls.
    alpha: (N, 1); K_y: the inverse of K + s I (:144)."""
    sqdist = ((a[:, :, None] - b[:, :, None].T) ** 2).sum(1)                  # :43
    sigma_grad = 2 * sigma * np.exp(-.5 * sqdist / (l ** 2))                 # :48
    sigma_matrix = np.dot(np.dot(alpha, alpha.T) - K_y, sigma_grad)          # :49
    sigma_var = .5 * np.diagonal(sigma_matrix).sum()                         # :50-51
    l_grad = sigma ** 2 * np.exp(-.5 * sqdist / (l ** 2)) * (sqdist / l ** 3)  # :54
    l_matrix = np.dot(np.dot(alpha, alpha.T) - K_y, l_grad)                  # :55
    l_var = .5 * np.diagonal(l_matrix).sum()                                 # :56-57
    return l_var, sigma_var


def gradient_ascent(a, b, sigma, l, alpha, K_y):
    """tune_hyperparms_regression.py:31-64: one ascent step on l (sigma is returned unchanged,
    its update is commented out at :61)."""
    l_var, _ = lml_gradient_terms(a, b, sigma, l, alpha, K_y)
    return sigma, l + GA_STEP_SIZE * l_var                                   # :63


def lml_and_gradient(X_train, y_train, sigma, l, s=NOISE_VAR):
    """One iteration of tune_hyperparms_first's loop body without the predictive part
    (tune_hyperparms_regression.py:123-145): LML, (l_var, sigma_var), alpha, K_y_inv."""
    n = len(X_train)
    K_train = RBF_kernel(X_train, X_train, sigma, l)                         # :123
    L = np.linalg.cholesky(K_train + s * np.eye(n))                          # :127
    m = np.linalg.solve(L, y_train)                                          # :128
    alpha = np.linalg.solve(L.T, m)                                          # :129
    lml = (-.5 * np.dot(y_train.T, alpha) - np.log(np.diagonal(L)).sum(0)
           - n / 2.0 * np.log(2 * np.pi))                                    # :141
    K_y_inv = np.dot(np.linalg.inv(L.T), np.linalg.inv(L))                   # :144
    l_var, sigma_var = lml_gradient_terms(X_train, X_train, sigma, l, alpha.reshape(-1, 1), K_y_inv)
    return lml, l_var, sigma_var, alpha, K_y_inv


def tune_hyperparms_first(X_train, X_test, y_train, num_fun, sigma, l, max_iter=GA_MAX_ITER):
    """tune_hyperparms_regression.py:104-162 (prints and the plt.axis call dropped).  Returns
    (mu_post, stand_devi, f_post_fun, optimal_likelihood) and, as extras for the tests,
    the final l and the iteration count."""
    s = NOISE_VAR                                                            # :115
    log_marg_likelihood_old = 0                                              # :116
    N = len(X_test)
    it = 0
    for i in range(max_iter):                                                # :121
        it = i + 1
        p = posterior(X_train, X_test, y_train, sigma, l, s)                 # :123-138
        with np.errstate(invalid='ignore'):
            stand_devi = np.sqrt(p['var'])
        n = len(X_train)
        log_marg_likelihood = (-.5 * np.dot(y_train.T, p['alpha']) - np.log(np.diagonal(p['L'])).sum(0)
                               - n / 2.0 * np.log(2 * np.pi))                # :141
        K_y_inv = np.dot(np.linalg.inv(p['L'].T), np.linalg.inv(p['L']))     # :144
        sigma, l = gradient_ascent(X_train, X_train, sigma, l, p['alpha'].reshape(-1, 1), K_y_inv)  # :145
        error = np.sqrt(np.sum((log_marg_likelihood - log_marg_likelihood_old) ** 2))  # :147
        log_marg_likelihood_old = log_marg_likelihood                        # :148
        if error <= GA_TOLERANCE:                                            # :149
            break
    L_ = np.linalg.cholesky(p['K_ss'] + POST_JITTER * np.eye(N) - np.dot(p['v'].T, p['v']))   # :159
    f_post_fun = p['mu'].reshape(-1, 1) + np.dot(L_, np.random.normal(size=(N, num_fun)))     # :160
    return p['mu'], stand_devi, f_post_fun, log_marg_likelihood, l, it


# ---------------------------------------------------------------------------
# SURVEY.md section 8f row f4 (second half): the composite covariance of CO2_example.py
# (Python-2 file, not importable; restated statement by statement)
# ---------------------------------------------------------------------------
def co2_covariance_function(a, b, hyperparms):
    """CO2_example.py:66-90 with kernel_1..kernel_4 (:9-64) inlined in the same order."""
    if a.shape[1] == 1 and b.shape[1] == 1:                                            # :75
        sqdist = ((a[:, :, None] - b[:, :, None].T) ** 2).sum(1)                       # :76
        l2_norm = np.sqrt(sqdist)                                                      # :77
    else:
        sqdist = (((a[:, None, :] - b[None, :, :]) ** 2).sum(axis=2))                  # :83
        l2_norm = np.sqrt(sqdist)                                                      # :85
    t = hyperparms
    k1 = (t[0] ** 2) * np.exp(-.5 * sqdist / t[1] ** 2)                                # :17
    first_item = -.5 * sqdist / t[3] ** 2                                              # :30
    second_item = -2 * ((np.sin(np.pi * l2_norm)) / t[4]) ** 2                         # :31
    k2 = t[2] ** 2 * np.exp(first_item + second_item)                                  # :32
    item = 1 + .5 * sqdist / (t[7] * t[6] ** 2)                                        # :44
    k3 = t[5] ** 2 * (1.0 / np.power(item, t[7]))                                      # :45-46
    n = len(sqdist)
    delta = np.eye(n) if sqdist.shape[0] == sqdist.shape[1] else 0                     # :58-62
    k4 = t[8] ** 2 * np.exp(-.5 * sqdist / t[9] ** 2) + t[10] ** 2 * delta             # :63-64
    return k1 + k2 + k3 + k4                                                           # :86-89


def co2_compute_mar_likelihood(X_train, y_train, hyperparms):
    """CO2_example.py:125-142 (alpha through the explicit inverse of L, as there)."""
    s = NOISE_VAR                                                                      # :133
    n = len(X_train)
    K_train = co2_covariance_function(X_train, X_train, hyperparms)                    # :135
    L = np.linalg.cholesky(K_train + s * np.eye(n))                                    # :136
    L_inv = np.linalg.inv(L)                                                           # :137
    alpha = np.dot(L_inv.T, np.dot(L_inv, y_train))                                    # :138
    return (-.5 * np.dot(y_train.T, alpha) - np.log(np.diagonal(L)).sum(0)
            - n / 2.0 * np.log(2 * np.pi))                                             # :140


def co2_posterior(X_train, X_test, y_train, hyperparms, s):
    """The body shared by bayesian_opt (:155-171) and make_prediction (:183-198)."""
    n = len(X_train)
    K = co2_covariance_function(X_train, X_train, hyperparms)
    K_s = co2_covariance_function(X_train, X_test, hyperparms)
    K_ss = co2_covariance_function(X_test, X_test, hyperparms)
    L = np.linalg.cholesky(K + s * np.eye(n))
    L_inv = np.linalg.inv(L)
    alpha = np.dot(L_inv.T, np.dot(L_inv, y_train))
    mu_post = np.dot(K_s.T, alpha)
    v = np.dot(L_inv, K_s)
    var_test = np.diag(K_ss) - np.sum(v ** 2, axis=0)
    with np.errstate(invalid='ignore'):
        stand_devi = np.sqrt(var_test)
    return mu_post, stand_devi, K_ss, v


def co2_bayesian_opt(hyperparms_train, hyperparms_test, y_train):
    """CO2_example.py:145-172: GP over hyper-parameter vectors with the composite kernel whose own
    hyper-parameters are the first training vector (:157); s = 1e-4; returns (mu, sd)."""
    mu, sd, _, _ = co2_posterior(hyperparms_train, hyperparms_test, y_train, hyperparms_train[0], BO_NOISE_VAR)
    return mu, sd


def co2_make_prediction(X_train, X_test, y_train, hyperparms):
    """CO2_example.py:175-203: s = 5e-4, one posterior sample."""
    N = len(X_test)
    mu, sd, K_ss, v = co2_posterior(X_train, X_test, y_train, hyperparms, NOISE_VAR)
    L_ = np.linalg.cholesky(K_ss + POST_JITTER * np.eye(N) - np.dot(v.T, v))           # :201
    f_post_fun = mu.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(N, 1)))         # :202
    return mu, sd, f_post_fun


# ---------------------------------------------------------------------------
# Memory-feasible restatement: identical K arithmetic, but true
# triangular solves instead of LU on a triangular matrix.  Used as the CPU
# baseline at sizes where the (N,d,N) broadcast does not fit, and as the
# checker at mid sizes (N of a few thousand) where LU would take minutes.
# ---------------------------------------------------------------------------
def fit_predict_feasible(X_train, X_test, y_train, sigma, l, s, rows=256, use_c=True, timings=None):
    """timings: optional dict that receives the wall seconds of each stage (bench.py's cpu_baseline:
    kbuild, chol, trsv (both vector solves), ks, trsm, meanvar)."""
    import time
    import scipy.linalg as sla
    N = len(X_train)
    if use_c:
        rbf = lambda a, b, sg, ll, rows=None: RBF_kernel_c(a, b, sg, ll)  # noqa: E731
    else:
        rbf = RBF_kernel_chunked
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + (t[-1] - t[-2])
    K = rbf(X_train, X_train, sigma, l, rows)
    K[np.diag_indices(N)] += s
    lap("kbuild")
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    lap("chol")
    m = sla.solve_triangular(L, y_train, lower=True, check_finite=False)
    alpha = sla.solve_triangular(L, m, lower=True, trans='T', check_finite=False)
    lap("trsv")
    K_s = rbf(X_train, X_test, sigma, l, rows)
    mu = K_s.T @ alpha
    lap("ks")
    v = sla.solve_triangular(L, K_s, lower=True, overwrite_b=True, check_finite=False)
    lap("trsm")
    var = sigma ** 2 - np.einsum('ij,ij->j', v, v)
    lml = (-.5 * float(y_train @ alpha) - float(np.log(np.diagonal(L)).sum())
           - N / 2.0 * np.log(2 * np.pi))
    lap("meanvar")
    return dict(mu=mu, var=var, alpha=alpha, m=m, lml=lml, diagL=np.diagonal(L).copy())


def fit_predict_blocked(X_train, X_test, y_train, sigma, l, s, block=4096, timings=None, progress=None):
    """fit_predict_feasible for sizes where LAPACK's dpotrf on the host would take too long (N = 131072: 7.5e14 flop):
    the same statements (GP_regression.py:138-148, tune_hyperparms_regression.py:312), the Cholesky factorisation
    written out as the blocked right-looking algorithm LAPACK itself uses -- dpotrf on a diagonal block, dtrsm on the
    rows below, dsyrk / dgemm on the trailing lower block columns -- with a block of 4096 instead of LAPACK's 64-256, so
    that nearly all flops are large dgemm calls.  Same arithmetic per element up to the order of the K-long sums; checked
    against fit_predict_feasible in tests/test_oracle_vs_golden.py.  In place on the lower triangle of K."""
    import time
    import scipy.linalg as sla
    N = len(X_train)
    t = [time.perf_counter()]

    def lap(name):
        t.append(time.perf_counter())
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + (t[-1] - t[-2])
    A = RBF_kernel_c(X_train, X_train, sigma, l)
    A[np.diag_indices(N)] += s                                                   # :138
    lap("kbuild")
    for k in range(0, N, block):
        e = min(k + block, N)
        Lkk = sla.cholesky(A[k:e, k:e], lower=True, check_finite=False)          # dpotrf
        A[k:e, k:e] = Lkk
        if e < N:
            P = sla.solve_triangular(Lkk, A[e:, k:e].T, lower=True, check_finite=False).T   # dtrsm: rows below
            A[e:, k:e] = P
            for j in range(e, N, block):                                         # trailing update, lower block columns only
                je = min(j + block, N)
                A[j:, j:je] -= P[j - e:] @ P[j - e:je - e].T
        if progress is not None:
            progress(e, N)
    lap("chol")
    diagL = np.diagonal(A).copy()
    # triangular solves against the lower triangle (the strict upper triangle still holds K: masked by `lower=True`)
    m = sla.solve_triangular(A, y_train, lower=True, check_finite=False)         # :139
    alpha = sla.solve_triangular(A, m, lower=True, trans='T', check_finite=False)    # :140
    lap("trsv")
    K_s = RBF_kernel_c(X_train, X_test, sigma, l)
    mu = K_s.T @ alpha                                                           # :143
    lap("ks")
    v = sla.solve_triangular(A, K_s, lower=True, overwrite_b=True, check_finite=False)   # :144
    lap("trsm")
    var = sigma ** 2 - np.einsum('ij,ij->j', v, v)                               # :147
    lml = (-.5 * float(y_train @ alpha) - float(np.log(diagL).sum()) - N / 2.0 * np.log(2 * np.pi))   # tune...:312
    lap("meanvar")
    return dict(mu=mu, var=var, alpha=alpha, m=m, lml=lml, diagL=diagL)


def synthetic_problem(N, d, n, seed=20240531):
    """SURVEY.md section 8(d) synthetic inputs (the bench / parity workload)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
    Xs = rng.uniform(-1, 1, (n, d))
    return X, y, Xs
