"""Drop-in for the hot-path functions of the reference's GP_regression.py.

Same names, positional order, return shapes/dtypes and hard-coded constants as
/root/reference/GP_regression.py (s = 0.0005 and sigma = 1 inside `prediction`,
:120-121; jitter 1e-6, :154); the arithmetic runs on the MI355X through
libgpmi355x.so.  Only the squared-exponential kernel is in scope
(BASELINE.json north_star): 'lin' / 'per' raise.
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


def _rbf_only(kernel_choice):
    if kernel_choice != 'rbf':
        raise NotImplementedError(
            "kernel_choice=%r: only the squared-exponential ('rbf') path is implemented on the "
            "MI355X (linear / periodic kernels are out of scope, SURVEY.md section 8f row f4)" % (kernel_choice,))


def prediction(X_train, X_test, y_train, kernel_choice, l, num_fun, *, sigma=SIGMA_F,
               noise_var=NOISE_VAR, jitter=POST_JITTER, ctx=None):
    """GP posterior at the test points, reference GP_regression.py:109-156.

    :return: (mu_post (n,), stand_devi (n,), f_post_fun (n, num_fun))
    Raises numpy.linalg.LinAlgError where the reference's np.linalg.cholesky
    would (K + sI at :138, posterior covariance at :154).  The normals of :155
    are drawn on the host from np.random in the reference's order.
    """
    _rbf_only(kernel_choice)
    ctx = ctx or default_context()
    ctx.fit(X_train, y_train, sigma, l, noise_var)        # :126,138-140
    mu_post, stand_devi = ctx.predict(X_test, want_sd=True)  # :127,143-148
    n = mu_post.shape[0]
    L_ = ctx.post_chol(jitter)                            # :154
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(n, num_fun)))  # :155
    return mu_post, stand_devi, f_post_fun


def f_prior(X_test, mu_prior, kernel_choice, kernel_parameter, num_fun, *, ctx=None):
    """GP prior samples, reference GP_regression.py:71-92 (s = 0.0005, sigma = 1)."""
    _rbf_only(kernel_choice)
    ctx = ctx or default_context()
    X_test = np.asarray(X_test, dtype=np.float64)
    num_test = len(X_test)
    ctx.fit(X_test, np.zeros(num_test), SIGMA_F, kernel_parameter, NOISE_VAR)   # :90
    B = ctx.factor()
    return mu_prior + np.dot(B, np.random.normal(size=(num_test, num_fun)))     # :91
