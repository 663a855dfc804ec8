"""Drop-in for the GP path of the reference's CO2_example.py (a Python-2 script): the
composite covariance function and the fit / predict / log-marginal-likelihood it drives,
on the MI355X through the C-ABI (second half).

Same names, argument order and return values as the reference:
    covariance_function(a, b, hyperparms)                       CO2_example.py:66-90
    compute_mar_likelihood(X_train, y_train, hyperparms)        :125-142
    bayesian_opt(hyperparms_train, hyperparms_test, y_train)    :145-172
    make_prediction(X_train, X_test, y_train, hyperparms)       :175-203
and the host-side pieces of its Bayesian-optimisation loop (acquisition functions, candidate
sampling, :93-122, :206-306).  Out of scope: the Mauna Loa download (`fetch_mldata`, :405 -- a
network fetch of an API that no longer exists), plotting, and the printing loop body of
tune_hyperparameters_BO, which is restated without its prints and plots.

The reference gets alpha through an explicit inverse of L (:137-138); here it is the same two
triangular solves as everywhere else (mathematically identical, better conditioned).
"""
from __future__ import annotations

import random

import numpy as np
from scipy.stats import norm

from .gp import default_context
from .tune_hyperparms_regression import overlap

NOISE_VAR = 0.0005      # CO2_example.py:133, :183
BO_NOISE_VAR = 0.0001   # CO2_example.py:154
HYPERMS_BOOK = np.array([66, 67, 2.4, 90, 1.3, .66, 1.2, .78, .18, 1.6, .19])   # :120, :303, :323


def covariance_function(a, b, hyperparms, *, ctx=None):
    """kernel_1 + kernel_2 + kernel_3 + kernel_4 (:9-64) of the pairwise distances of a and b;
    kernel_4's theta_11^2 * eye is added whenever the result is square, as in the reference (:58)."""
    ctx = ctx or default_context()
    return ctx.cov("co2", a, b, hyperparms)


def _fit(ctx, X_train, y_train, hyperparms, s):
    """K + sI -> L, alpha with the composite kernel (:135-138); callers switch the context back
    to the squared-exponential kernel when they are done."""
    ctx.set_kernel("co2", hyperparms)
    return ctx.fit(X_train, y_train, 1.0, 1.0, s)


def compute_mar_likelihood(X_train, y_train, hyperparms, *, ctx=None):
    """Log marginal likelihood under the composite kernel, reference :125-142."""
    ctx = ctx or default_context()
    try:
        return np.float64(_fit(ctx, X_train, y_train, hyperparms, NOISE_VAR))
    finally:
        ctx.set_kernel("rbf")


def bayesian_opt(hyperparms_train, hyperparms_test, y_train, *, ctx=None):
    """GP over hyper-parameter vectors, reference :145-172: the kernel's own hyper-parameters
    are the first training vector (:157), s = 1e-4; returns (mu_post, stand_devi)."""
    ctx = ctx or default_context()
    hyperparms_train = np.asarray(hyperparms_train, dtype=np.float64)
    try:
        _fit(ctx, hyperparms_train, y_train, hyperparms_train[0], BO_NOISE_VAR)
        return ctx.predict(hyperparms_test, want_sd=True)
    finally:
        ctx.set_kernel("rbf")


def make_prediction(X_train, X_test, y_train, hyperparms, *, ctx=None):
    """Posterior mean, standard deviation and one posterior sample, reference :175-203."""
    ctx = ctx or default_context()
    try:
        _fit(ctx, X_train, y_train, hyperparms, NOISE_VAR)
        mu_post, stand_devi = ctx.predict(X_test, want_sd=True)
        N = mu_post.shape[0]
        num_fun = 1                                              # :199
        L_ = ctx.post_chol(1e-6)                                 # :201
    finally:
        ctx.set_kernel("rbf")
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(N, num_fun)))   # :202
    return mu_post, stand_devi, f_post_fun


# ---------------------------------------------------------------------------------------
# host side of the reference's BO loop (:93-122, :206-306), Python-3 restatement
# ---------------------------------------------------------------------------------------
def get_pos_data(info):
    """:93-102"""
    data = info[:, 1:13].flatten()
    neg_indices = np.where(data <= 0)
    return np.delete(data, neg_indices)


def random_sample_test_parms(n_test_hyperparms, train_parms):
    """:105-122 (random.sample over a list: Python 3 refuses an ndarray population)"""
    dim_parms = 11
    lower = HYPERMS_BOOK - HYPERMS_BOOK * .7
    upper = HYPERMS_BOOK + HYPERMS_BOOK * .5
    test_parms_matrix = np.zeros(shape=(n_test_hyperparms, dim_parms))
    for i in range(dim_parms):
        num_gen = n_test_hyperparms + len(train_parms) + 10
        test_parms = np.linspace(lower[i], upper[i], num_gen)
        ind_done, ind_sample = overlap(train_parms[:, i], test_parms)
        test_parms = np.delete(test_parms, ind_sample.astype(int))
        test_parms_matrix[:, i] = np.asarray(random.sample(list(test_parms), n_test_hyperparms))
    return test_parms_matrix


def UBC(hyperparms_train, hyperparms_test, mu_post, stand_devi):
    """Upper confidence bound, :206-224 (kappa = 7; True when the last point is proposed again)."""
    num_parms = len(hyperparms_train)
    kappa = 7
    objective = mu_post + kappa * stand_devi
    indices = np.asarray(np.where(objective == np.max(objective))[0])
    next_point = hyperparms_test[indices[0]]
    if np.array_equal(hyperparms_train[num_parms - 1], next_point):
        return True
    return next_point


def TS(hyperparms_train, hyperparms_test, y_train, *, ctx=None):
    """Thompson sampling, :227-238.  The reference unpacks three values from its two-valued
    bayesian_opt (a ValueError if the branch were ever reached); here the sample is drawn from
    make_prediction's posterior with the same kernel hyper-parameters."""
    hyperparms_train = np.asarray(hyperparms_train, dtype=np.float64)
    ctx = ctx or default_context()
    try:
        _fit(ctx, hyperparms_train, y_train, hyperparms_train[0], BO_NOISE_VAR)
        mu_post, _ = ctx.predict(hyperparms_test, want_sd=True)
        L_ = ctx.post_chol(1e-6)
    finally:
        ctx.set_kernel("rbf")
    f_post_fun = mu_post.reshape(-1, 1) + np.dot(L_, np.random.normal(size=(len(mu_post), 1)))
    max_index = np.where(f_post_fun == np.max(f_post_fun))
    return hyperparms_test[max_index[0]].flatten()


def EI(hyperparms_test, mu_post, stand_devi, y):
    """Expected improvement, :241-258."""
    s = 0.0005
    f_max = np.max(y) + s
    z = (mu_post - f_max) / stand_devi
    EI_vector = (mu_post - f_max) * norm.cdf(z) + stand_devi * norm.pdf(z)
    max_index = np.where(EI_vector == np.max(EI_vector))
    return hyperparms_test[max_index].flatten()


def PI(hyperparms_test, mu_post, stand_devi, y):
    """Probability of improvement, :261-280 (ties broken by random.randint, as there)."""
    s = 0.0005
    f_max = np.max(y) + s
    z = (mu_post - f_max) / stand_devi
    cumu_gaussian = norm.cdf(z)
    indices = np.asarray(np.where(cumu_gaussian == np.max(cumu_gaussian))[0])
    rand_index = random.randint(0, len(indices) - 1)
    return hyperparms_test[indices[rand_index]]


def acquisition_fun(choice, hyperparms_train, hyperparms_test, mu_post, stand_devi, y, *, ctx=None):
    """:283-300: 'UBC' / 'TS' / 'EI', anything else -> PI."""
    if choice == 'UBC':
        return UBC(hyperparms_train, hyperparms_test, mu_post, stand_devi)
    if choice == 'TS':
        return TS(hyperparms_train, hyperparms_test, y, ctx=ctx)
    if choice == 'EI':
        return EI(hyperparms_test, mu_post, stand_devi, y)
    return PI(hyperparms_test, mu_post, stand_devi, y)


def init_hyperms(n_hyperms, dim_parms):
    """:296-306"""
    hyperparms = np.zeros(shape=(n_hyperms, dim_parms))
    for i in range(n_hyperms):
        hyperparms[i] = HYPERMS_BOOK + 0.5 * (i + 5)
    return hyperparms


def tune_hyperparameters_BO(X_train, X_test, y_train, *, choices=('UBC', 'TS', 'EI', 'PI'), num_iterations=10,
                            n_hyperparms_test=500, ctx=None, return_trace=False):
    """The BO loop over the 11 hyper-parameters, reference :309-371, without its prints and plots.
    The reference passes the whole list `choice` to acquisition_fun (:343), so every pass
    falls through to PI; here each pass uses its own entry of `choices` -- pass choices=('PI',)*4
    for the reference's effective behaviour.  The LML of a hyper-parameter vector is evaluated
    once and cached (the reference recomputes all of them every iteration, :338-339).
    Returns the best hyper-parameter vector of the last pass (and the per-pass traces)."""
    ctx = ctx or default_context()
    dim_parms = 11
    n_train_hyperparms = 5
    traces = {}
    passes = []                     # per pass: the choice, the running maxima, the points the acquisition chose
    best = None
    for j in choices:
        hyperparms_train = init_hyperms(n_train_hyperparms, dim_parms)
        cache = {}
        y_axis = np.zeros(num_iterations)
        chosen = []
        for k in range(num_iterations):
            hyperparms_test = random_sample_test_parms(n_hyperparms_test, hyperparms_train)
            lml = np.zeros(len(hyperparms_train))
            for i, h in enumerate(hyperparms_train):                       # :338-339
                key = h.tobytes()
                if key not in cache:
                    cache[key] = compute_mar_likelihood(X_train, y_train, h, ctx=ctx)
                lml[i] = cache[key]
            mu_post, stand_devi = bayesian_opt(hyperparms_train, hyperparms_test, lml, ctx=ctx)   # :341
            next_point = acquisition_fun(j, hyperparms_train, hyperparms_test, mu_post, stand_devi, lml, ctx=ctx)
            y_axis[k] = np.max(lml)
            best = hyperparms_train[int(np.argmax(lml))]
            if next_point is True:
                break
            chosen.append(np.asarray(next_point, dtype=np.float64).reshape(-1))
            hyperparms_train = np.append(hyperparms_train, [chosen[-1]], axis=0)   # :344
        traces[j] = y_axis
        passes.append({"choice": j, "y_axis": y_axis, "chosen": np.array(chosen), "train": hyperparms_train})
    if return_trace == "passes":
        return best, passes
    if return_trace:
        return best, traces
    return best
