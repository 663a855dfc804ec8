"""Drop-in for the hot-path functions of the reference's
tune_hyperparms_regression.py (which is Python-2 syntax and cannot be imported):
`compute_mar_likelihood` (:292-313), the batch its callers loop over
(:368-369, :385-386) and `bayesian_opt` (:67-101).
"""
from __future__ import annotations

import numpy as np

from .gp import default_context

NOISE_VAR = 0.0005      # tune_hyperparms_regression.py:302
BO_NOISE_VAR = 0.0001   # tune_hyperparms_regression.py:75


def compute_mar_likelihood(X_train, X_test, y_train, sigma, l, *, noise_var=NOISE_VAR, ctx=None):
    """Log marginal likelihood, reference tune_hyperparms_regression.py:292-313.
    X_test is accepted and unused, exactly as in the reference."""
    ctx = ctx or default_context()
    return np.float64(ctx.fit(X_train, y_train, sigma, l, noise_var))


def compute_mar_likelihood_batch(X_train, y_train, triples, *, ctx=None):
    """The reference's `for i in range(len(l)): compute_mar_likelihood(...)` loops
    (:368-369, :385-386) as one call: triples is (T, 3) rows of
    (l, sigma_f, noise_var).  Returns lml (T,), NaN where K + sI was not PD."""
    ctx = ctx or default_context()
    ctx.set_train(X_train, y_train)
    lml, _ = ctx.lml_batch(triples)
    return lml


def bayesian_opt(X_train, X_test, y_train, *, ctx=None):
    """Surrogate GP of the Bayesian-optimisation loop, reference :67-101
    (s = 1e-4, sigma = l = 1, one posterior sample)."""
    ctx = ctx or default_context()
    ctx.fit(X_train, y_train, 1, 1, BO_NOISE_VAR)          # :80-87
    mu_post, stand_devi = ctx.predict(X_test, want_sd=True)  # :90-95
    n = mu_post.shape[0]
    L_ = ctx.post_chol(1e-6)                               # :98
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(n, 1)))  # :99
    return mu_post, stand_devi, f_post_fun
