"""Drop-in for the hot-path functions of the reference's GP_regression.py.

Same names, positional order, return shapes/dtypes and hard-coded constants as
/root/reference/GP_regression.py (s = 0.0005 and sigma = 1 inside `prediction`,
:120-121; jitter 1e-6, :154); the arithmetic runs on the MI355X through
libgpmi355x.so.  The squared-exponential kernel is the optimised path
(BASELINE.json north_star); the reference's linear and periodic kernels
 run through the same factorisation with a plain
kernel-matrix build.
"""
from __future__ import annotations

import numpy as np

from .gp import default_context

NOISE_VAR = 0.0005      # GP_regression.py:120
SIGMA_F = 1             # GP_regression.py:121
POST_JITTER = 1e-6      # GP_regression.py:154


def RBF_kernel(a, b, sigma, l):
    """RBF kernel matrix, reference GP_regression.py:8-19.

    :param a: (N, d) inputs
    :param b: (M, d) inputs
    :param sigma: output scale (the kernel is sigma**2 * exp(...))
    :param l: lengthscale (scalar or 1-element array)
    :return: (N, M) float64 covariance matrix
    """
    return default_context().rbf(a, b, sigma, l)


def dataset_generator(N, n):
    """Synthetic 1-D sine data, reference GP_regression.py:53-68.  Host-side and
    NumPy-RNG-order compatible (uniform then randn) so seeded runs reproduce."""
    s = 0.0005
    f = lambda x: np.sin(0.9 * x).flatten()  # noqa: E731
    X_train = np.random.uniform(-5, 5, size=(N, 1))
    y_train = f(X_train) + np.sqrt(s) * np.random.randn(N)
    X_test = np.linspace(-5, 5, n).reshape(-1, 1)
    return f, X_train, y_train, X_test


def lin_kernel(a, b, c):
    """linear kernel np.dot(a - c, b.T - c), reference GP_regression.py:22-33"""
    return default_context().cov('lin', a, b, c)


def per_kernel(a, b, parameters):
    """periodic kernel exp(-2 sin(pi |a-b| / p)^2 / l^2) on 1-D inputs, parameters = (p, l);
    reference GP_regression.py:36-50"""
    p, l = parameters
    return default_context().cov('per', a, b, p, l)


def _select_kernel(ctx, kernel_choice, parameter, sigma):
    """kernel_choice / parameter conventions of prediction() (GP_regression.py:125-136):
    'rbf': l; 'lin': the offset c; 'per': the tuple (p, l)."""
    if kernel_choice == 'rbf':
        ctx.set_kernel('rbf')
        return sigma, parameter
    if kernel_choice == 'lin':
        ctx.set_kernel('lin', parameter)
        return 1.0, 1.0
    if kernel_choice == 'per':
        p, l = parameter
        ctx.set_kernel('per', p, l)
        return 1.0, 1.0
    raise ValueError("kernel_choice must be 'rbf', 'lin' or 'per', got %r" % (kernel_choice,))


def _dist_of(n_gpus, dist):
    """n_gpus / dist keywords: None -> single-GPU context; a DistGP -> that; n_gpus > 1 -> the
    process-wide DistGP over WORLD (every rank must make the same call)."""
    if dist is not None:
        return dist
    if n_gpus is not None and int(n_gpus) > 1:
        from .dist import default_dist
        return default_dist(n_gpus)
    return None


def prediction(X_train, X_test, y_train, kernel_choice, l, num_fun, *, sigma=SIGMA_F,
               noise_var=NOISE_VAR, jitter=POST_JITTER, return_lml=False, ctx=None, n_gpus=None, dist=None):
    """GP posterior at the test points, reference GP_regression.py:109-156.

    :return: (mu_post (n,), stand_devi (n,), f_post_fun (n, num_fun)); with return_lml=True a
             fourth element, the log marginal likelihood of the fit (tune_hyperparms_regression.py:312
             -- not the discarded expression at GP_regression.py:151, which lacks the log)
    Raises numpy.linalg.LinAlgError where the reference's np.linalg.cholesky
    would (K + sI at :138, posterior covariance at :154).  The normals of :155
    are drawn on the host from np.random in the reference's order.
    """
    gp = _dist_of(n_gpus, dist)
    if gp is not None:
        # the covariance row-block partitioned over the ranks of the node (dist.py); every rank makes this call
        # with the same arguments and gets the same results; the normals of :155 come from each rank's own
        # np.random (seed them alike for identical samples)
        try:
            sg, ll = _select_kernel(gp, kernel_choice, l, sigma)      # :125-136, the same three choices on every path
            from ._lib import scalar
            lml = gp.fit(X_train, y_train, sg, scalar(ll, "l"), noise_var)
            mu_post, stand_devi = gp.predict(X_test, want_sd=True)
            L_ = gp.post_chol(jitter)
        finally:
            gp.set_kernel('rbf')
        f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(mu_post.shape[0], num_fun)))
        if return_lml:
            return mu_post, stand_devi, f_post_fun, np.float64(lml)
        return mu_post, stand_devi, f_post_fun
    ctx = ctx or default_context()
    try:
        sg, ll = _select_kernel(ctx, kernel_choice, l, sigma)  # :125-136
        # :126-128, 138-148 and 153-154 in one pass: ONE Cholesky of [[K + sI, .], [K(X*, X), K_ss + jitter I]] -- the test
        # rows ride through it below the training rows, and its last n columns are L_ (gpmi_fit_predict_sample_resident)
        lml, mu_post, stand_devi, _ = ctx.fit_predict_sample(X_train, y_train, X_test, sg, ll, noise_var, jitter, want_sd=True,
                                                             want_factor=False)
        n = mu_post.shape[0]
        normals = np.random.normal(size=(n, num_fun))          # :155, drawn on the host in the reference's order
        LZ = ctx.post_sample(jitter, normals)                  # L_ stays on the device: n x num_fun crosses PCIe, not n x n
    finally:
        ctx.set_kernel('rbf')
    f_post_fun = mu_post.reshape(-1, 1) + LZ               # :155
    if return_lml:
        return mu_post, stand_devi, f_post_fun, np.float64(lml)
    return mu_post, stand_devi, f_post_fun


def f_prior(X_test, mu_prior, kernel_choice, kernel_parameter, num_fun, *, ctx=None):
    """GP prior samples, reference GP_regression.py:71-92 (s = 0.0005, sigma = 1)."""
    ctx = ctx or default_context()
    X_test = np.asarray(X_test, dtype=np.float64)
    num_test = len(X_test)
    try:
        sg, ll = _select_kernel(ctx, kernel_choice, kernel_parameter, SIGMA_F)      # :84-89
        ctx.fit(X_test, np.zeros(num_test), sg, ll, NOISE_VAR)                      # :90
        B = ctx.factor()
    finally:
        ctx.set_kernel('rbf')
    return mu_prior + np.dot(B, np.random.normal(size=(num_test, num_fun)))     # :91
