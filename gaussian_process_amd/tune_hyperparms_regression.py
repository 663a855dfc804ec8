"""Drop-in for the hot-path functions of the reference's
tune_hyperparms_regression.py (which is Python-2 syntax and cannot be imported):
`compute_mar_likelihood` (:292-313), the batch its callers loop over
(:368-369, :385-386), `bayesian_opt` (:67-101), the Bayesian-optimisation loop
(:165-289, :316-395, :418-432) and the gradient-ascent tuner (:31-64, :104-162, :398-415).
"""
from __future__ import annotations

import numpy as np

from .gp import default_context

NOISE_VAR = 0.0005      # tune_hyperparms_regression.py:302
BO_NOISE_VAR = 0.0001   # tune_hyperparms_regression.py:75


def compute_mar_likelihood(X_train, X_test, y_train, sigma, l, *, noise_var=NOISE_VAR, ctx=None, n_gpus=None,
                           dist=None):
    """Log marginal likelihood, reference tune_hyperparms_regression.py:292-313.
    X_test is accepted and unused, exactly as in the reference.  n_gpus / dist: factorise with the covariance
    row-block partitioned over the ranks of the node (every rank makes the same call)."""
    from .GP_regression import _dist_of
    gp = _dist_of(n_gpus, dist)
    if gp is not None:
        from ._lib import scalar
        return np.float64(gp.fit(X_train, y_train, scalar(sigma, "sigma"), scalar(l, "l"), noise_var))
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


# ---------------------------------------------------------------------------------------
# the Bayesian-optimisation loop around the batched LML -- a
# Python-3 restatement of tune_hyperparms_regression.py:165-289 (acquisition functions) and
# :316-395, :418-432 (candidate sampling, the loop, the driver).  Host code: the surrogate
# GP has at most ~5 points; the accelerated part is compute_mar_likelihood_batch.
# Plotting (plot_BO, the plt calls inside the acquisition functions) is out of scope.
# Pinned: the reference's own functions, print statements passed through lib2to3's fixers, were run with `random`
# and `np.random` seeded (oracle/make_golden.py:bo_loop_cases); every candidate set, surrogate posterior, chosen
# lengthscale and returned maximum of those runs is in tests/golden/kernels_bo_loops.npz, and the loops below
# reproduce them point for point (tests/test_host_logic.py on the CPU, the -m gpu tests free-running).
# ---------------------------------------------------------------------------------------
import random  # noqa: E402

from scipy.stats import norm  # noqa: E402

from .GP_regression import prediction  # noqa: E402


def _is_done(point, parms_done):
    return float(np.ravel(point)[0]) in [float(v) for v in np.ravel(parms_done)]


def PI(params, means, stand_devi, parms_done, y, n_iterations, k):
    """Probability of improvement, reference :165-204.  Returns the next point (row of
    params) or True when the early-stop criteria fire."""
    s = 0.0005
    stop_threshold = 0.001
    f_max = np.max(y) + s
    cumu_gaussian = norm.cdf((means - f_max) / stand_devi)
    if cumu_gaussian.sum() <= stop_threshold or np.max(cumu_gaussian) <= stop_threshold:
        return True                                              # :179-181
    indices = np.where(cumu_gaussian == np.max(cumu_gaussian))[0]
    next_point = params[indices[random.randint(0, len(indices) - 1)]]
    if _is_done(next_point, parms_done):                        # :189-194: one redraw, then stop
        next_point = params[indices[random.randint(0, len(indices) - 1)]]
        if _is_done(next_point, parms_done):
            return True
    return next_point


def UCB(parms_done, params, means, stand_devi, n_iterations, k):
    """Upper confidence bound, reference :207-230 (kappa = 0.001)."""
    kappa = 0.001
    objective = means + kappa * stand_devi
    next_point = params[np.where(objective == np.max(objective))[0][0]]
    if np.ravel(parms_done)[-1] == np.ravel(next_point)[0]:
        return True
    return next_point


def TS(parms_done, params, y, n_iterations, k, ctx=None):
    """Thompson sampling, reference :233-250: one posterior sample of the surrogate."""
    mu_post, stand_devi, f_post_fun = prediction(np.asarray(parms_done, dtype=np.float64).reshape(-1, 1),
                                                 params, y, 'rbf', 1, 1, ctx=ctx)
    return params[np.where(f_post_fun == np.max(f_post_fun))]


def EI(params, means, stand_devi, parms_done, y, n_iterations, k):
    """Expected improvement, reference :253-273."""
    s = 0.0005
    f_max = np.max(y) + s
    z = (means - f_max) / stand_devi
    EI_vector = (means - f_max) * norm.cdf(z) + stand_devi * norm.pdf(z)
    return params[np.where(EI_vector == np.max(EI_vector))]


def acquisition_fun(params, means, stand_devi, parms_done, y, n_iterations, k, ctx=None):
    """Reference :275-289: all four are evaluated (they consume the RNG streams in this
    order), the PI point is the one returned (:289)."""
    next_point_PI = PI(params, means, stand_devi, parms_done, y, n_iterations, k)
    UCB(parms_done, params, means, stand_devi, n_iterations, k)
    TS(parms_done, params, y, n_iterations, k, ctx=ctx)
    EI(params, means, stand_devi, parms_done, y, n_iterations, k)
    return next_point_PI


def overlap(a, b):
    """Indices of the elements of a that occur in b and where they sit in b (both unique),
    reference :316-328."""
    ind_a = np.arange(len(a))[np.isin(a, b)]
    ind_b = np.array([np.argwhere(b == a[x]) for x in ind_a]).flatten()
    return ind_a, ind_b


def random_gen_test_parms(n, parms_done):
    """n sorted candidate lengthscales from linspace(0.01, 5, n + len(done) + 10) without the
    ones already evaluated, shape (n, 1); reference :331-346."""
    num_gen = n + len(parms_done) + 10
    test_parms = np.linspace(0.01, 5, num_gen)
    _, ind_sample = overlap(np.asarray(parms_done), test_parms)
    test_parms = np.delete(test_parms, ind_sample.astype(int))
    sampled = np.asarray(random.sample(list(test_parms), n))
    return np.sort(sampled).reshape(-1, 1)


def tune_hyperparms_second(X_train, X_test, y_train, num_fun, sigma, l, *, ctx=None, verbose=False,
                           return_trace=False):
    """The BO loop, reference :349-395: 3 iterations of {LML at every lengthscale tried so
    far (one batched call here), surrogate GP, PI acquisition}; returns the best LML."""
    ctx = ctx or default_context()
    n = 100
    n_iterations = 3
    l = np.asarray(l, dtype=np.float64).reshape(-1)

    def lml_of(ls):
        triples = np.column_stack([ls, np.full(len(ls), float(sigma)), np.full(len(ls), NOISE_VAR)])
        return compute_mar_likelihood_batch(X_train, y_train, triples, ctx=ctx)

    k = 0
    for k in range(n_iterations):
        l_test = random_gen_test_parms(n, l)
        log_marg_likelihood = lml_of(l)                                          # :368-369
        mu_post, stand_devi, _ = bayesian_opt(l.reshape(-1, 1), l_test, log_marg_likelihood, ctx=ctx)   # :371
        next_point = acquisition_fun(l_test, mu_post, stand_devi, l, log_marg_likelihood, n_iterations, k,
                                     ctx=ctx)                                    # :373
        if next_point is True:                                                   # :376-380
            break
        l = np.append(l, next_point)
    log_marg_likelihood = lml_of(l)                                              # :384-386
    best = int(np.argmax(log_marg_likelihood))
    if verbose:
        print("it takes %d iterations to get the optimal!" % (k + 1))
        print("optimal lenghscalar is: %r" % l[best])
        print("maximum likelihood is: %r" % np.max(log_marg_likelihood))
    if return_trace:
        return np.max(log_marg_likelihood), l, log_marg_likelihood
    return np.max(log_marg_likelihood)


def tune_hyperparms_BO(X_train, X_test, y_train, num_fun, *, ctx=None, verbose=False):
    """Reference :418-432: sigma = 1, two random initial lengthscales in [0.02, 5)."""
    sigma = 1
    l = np.random.uniform(0.02, 5, 2)
    return tune_hyperparms_second(X_train, X_test, y_train, num_fun, sigma, l, ctx=ctx, verbose=verbose)


# ---------------------------------------------------------------------------------------
# the gradient-ascent tuner, reference :31-64, :104-162, :398-415.
# The reference inverts L twice per iteration (:144) and multiplies two N x N matrices to read
# off a trace (:55); here the inverse comes from the resident factor on the device and the
# trace is one fused pass (gpmi_lml_grad), or -- for callers that hold alpha and K_y^-1
# themselves, as gradient_ascent's signature has it -- gpmi_grad_trace.
# ---------------------------------------------------------------------------------------
GA_STEP_SIZE = 0.01      # :42
GA_TOLERANCE = 0.001     # :117
GA_MAX_ITER = 10000      # :121


def gradient_ascent(a, b, sigma, l, alpha, K_y, *, ctx=None):
    """One ascent step on the lengthscale, reference :31-64 (same arguments: alpha is
    K_y^-1 y as a column, K_y the INVERSE of K + sI).  Returns (sigma, l): sigma unchanged,
    as its update is commented out in the reference (:61)."""
    ctx = ctx or default_context()
    l_var, _ = ctx.grad_trace(a, b, sigma, l, alpha, K_y)        # :43-57
    return sigma, l + GA_STEP_SIZE * l_var                       # :63


def lml_and_gradient(X_train, y_train, sigma, l, *, noise_var=NOISE_VAR, ctx=None):
    """LML (:141) and (dLML/dl, dLML/dsigma) (:54-57, :46-51) at (sigma, l) with everything
    resident on the device: the body of the tuner's loop without its predictive part."""
    ctx = ctx or default_context()
    lml = ctx.fit(X_train, y_train, sigma, l, noise_var)
    dl, ds = ctx.lml_grad()
    return np.float64(lml), dl, ds


def tune_hyperparms_first(X_train, X_test, y_train, num_fun, sigma, l, *, ctx=None, verbose=False,
                          max_iter=GA_MAX_ITER, return_trace=False):
    """Maximise the log marginal likelihood over the lengthscale by gradient ascent,
    reference :104-162.  Same constants (s = 0.0005, step 0.01, tolerance 1e-3 on |dLML|,
    at most 10000 iterations) and the same return values; the predictive mean / sd the
    reference recomputes every iteration (:131-138) are evaluated once, for the iterate the
    loop stops at -- which is what the reference returns."""
    ctx = ctx or default_context()
    ctx.set_train(X_train, y_train)
    s = NOISE_VAR                                                # :115
    log_marg_likelihood_old = 0                                  # :116
    l_eval = l
    it = 0
    for i in range(max_iter):                                    # :121
        it = i + 1
        l_eval = l
        log_marg_likelihood = np.float64(ctx.factorize(sigma, l, s))   # :123-129, :141
        l_var, _ = ctx.lml_grad()                                # :144-145 (:43-57)
        l = l + GA_STEP_SIZE * l_var                             # :63
        error = np.sqrt(np.sum((log_marg_likelihood - log_marg_likelihood_old) ** 2))   # :147
        log_marg_likelihood_old = log_marg_likelihood            # :148
        if error <= GA_TOLERANCE:                                # :149
            break
    if verbose:
        print("The hyperparameter tuning function has already converged after %d iterations!" % it)
        print("optimal lenghscalar is: %r" % float(np.asarray(l).reshape(-1)[0]))
        print("maximum log marginal likelihood is: %r" % log_marg_likelihood)
    # the factor resident now belongs to l_eval, the lengthscale of the last loop body
    mu_post, stand_devi = ctx.predict(X_test, want_sd=True)      # :131-138
    N = mu_post.shape[0]
    L_ = ctx.post_chol(1e-6)                                     # :159
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(N, num_fun)))   # :160
    if return_trace:
        return mu_post, stand_devi, f_post_fun, log_marg_likelihood, l, it
    return mu_post, stand_devi, f_post_fun, log_marg_likelihood


def tune_hyperparms_gradient(X_train, X_test, y_train, num_fun, *, ctx=None, verbose=False):
    """Reference :398-415: sigma = 1 and a random initial lengthscale in [0, 5); returns the
    maximal log marginal likelihood (the reference's plotting calls are out of scope)."""
    sigma = 1                                                    # :407
    l = np.random.uniform(0, 5, 1)                               # :408
    _, _, _, optimal_likelihood = tune_hyperparms_first(X_train, X_test, y_train, num_fun, sigma, l,
                                                        ctx=ctx, verbose=verbose)   # :410
    return optimal_likelihood
