"""Host-side shim logic that needs no GPU: argument coercion, error mapping, constants."""
import numpy as np
import pytest

from gaussian_process_amd import _lib
from gaussian_process_amd import GP_regression as G
from gaussian_process_amd import tune_hyperparms_regression as T


def test_scalar_accepts_what_the_reference_passes():
    # l[i] at tune_hyperparms_regression.py:369, np.random.uniform(0,5,1) at :408
    assert _lib.scalar(np.array([0.7])) == 0.7
    assert _lib.scalar(np.array(2.5)) == 2.5
    assert _lib.scalar(3) == 3.0
    with pytest.raises(ValueError):
        _lib.scalar(np.array([1.0, 2.0]))


def test_as_f64_makes_contiguous_float64():
    a = np.arange(12, dtype=np.int32).reshape(3, 4)[:, ::2]
    b = _lib.as_f64(a, 2)
    assert b.dtype == np.float64 and b.flags.c_contiguous and b.shape == (3, 2)
    with pytest.raises(ValueError):
        _lib.as_f64(np.zeros(3), 2)


def test_status_mapping_matches_reference_exceptions():
    _lib.check(_lib.GPMI_OK)
    with pytest.raises(np.linalg.LinAlgError) as ei:      # np.linalg.cholesky, GP_regression.py:138
        _lib.check(_lib.GPMI_ERR_NOT_PD, bad_pivot=7)
    assert ei.value.bad_pivot == 7
    with pytest.raises(ValueError):
        _lib.check(_lib.GPMI_ERR_BAD_ARG)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.GPMI_ERR_RUNTIME)


def test_constants_are_the_reference_literals():
    assert G.NOISE_VAR == 0.0005 and G.SIGMA_F == 1 and G.POST_JITTER == 1e-6   # GP_regression.py:120-121,154
    assert T.NOISE_VAR == 0.0005 and T.BO_NOISE_VAR == 0.0001                    # tune...:302, :75


def test_unknown_kernel_choice_is_rejected_before_any_gpu_work():
    class Dummy:            # no GPU here: the choice must be validated on the host
        def set_kernel(self, *a):
            pass
    with pytest.raises(ValueError):
        G.prediction(np.zeros((2, 1)), np.zeros((2, 1)), np.zeros(2), "matern", 1.0, 1, ctx=Dummy())


def test_dataset_generator_follows_reference_rng_order(oracle):
    np.random.seed(3)
    f, X, y, Xs = G.dataset_generator(40, 9)
    np.random.seed(3)
    f2, X2, y2, Xs2 = oracle.dataset_generator(40, 9)
    assert np.array_equal(X, X2) and np.array_equal(y, y2) and np.array_equal(Xs, Xs2)
    assert X.shape == (40, 1) and y.shape == (40,) and Xs.shape == (9, 1)


# ---- pieces of the Bayesian-optimisation loop (host code) -------------
def test_acquisition_functions_follow_their_formulas():
    from scipy.stats import norm
    import random
    params = np.linspace(0.01, 5, 50).reshape(-1, 1)
    means = -(params[:, 0] - 2.0) ** 2
    sd = 0.1 + 0.05 * params[:, 0]
    done = np.array([0.5, 3.5])
    y = np.array([-2.0, -2.5])
    random.seed(0)
    nxt = T.PI(params, means, sd, done, y, 3, 0)
    cdf = norm.cdf((means - (y.max() + 0.0005)) / sd)
    assert nxt.shape == (1,) and cdf[np.where(params[:, 0] == nxt[0])[0][0]] == cdf.max()
    assert T.PI(params, np.full(50, -100.0), sd, done, y, 3, 0) is True           # nothing can improve: stop
    assert T.UCB(done, params, means, sd, 3, 0)[0] == params[np.argmax(means + 0.001 * sd), 0]
    assert T.UCB(np.array([0.5, params[np.argmax(means + 0.001 * sd), 0]]), params, means, sd, 3, 0) is True
    z = (means - (y.max() + 0.0005)) / sd
    ei = (means - (y.max() + 0.0005)) * norm.cdf(z) + sd * norm.pdf(z)
    assert T.EI(params, means, sd, done, y, 3, 0)[0] == params[np.argmax(ei), 0]


def test_candidate_sampling():
    import random
    random.seed(1)
    done = np.array([0.01, 5.0, 1.234])
    cand = T.random_gen_test_parms(100, done)
    assert cand.shape == (100, 1) and np.all(np.diff(cand[:, 0]) > 0)
    assert not np.isin(cand[:, 0], done).any()
    assert np.all((cand >= 0.01) & (cand <= 5.0))
    ia, ib = T.overlap(np.array([3.0, 7.0, 9.0]), np.array([9.0, 1.0, 3.0]))
    assert ia.tolist() == [0, 2] and ib.tolist() == [2, 0]


def test_acquisition_functions_vs_reference_source():
    """UCB, EI and overlap against vectors produced by executing the reference's own function
    source (tune_hyperparms_regression.py:207-230, :253-273, :316-328; oracle/make_golden.py)."""
    from conftest import golden
    g = golden("kernels_acq")
    done, y, params, mu, sd = g["done"], g["y"], g["params"], g["mu"], g["sd"]
    assert np.array_equal(np.asarray(T.UCB(done, params, mu, sd, 3, 0)), g["ucb"])
    assert np.array_equal(np.asarray(T.EI(params, mu, sd, done, y, 3, 0)), g["ei"])
    j = int(np.argmax(mu + 0.001 * sd))
    assert (T.UCB(np.append(done, params[j, 0]), params, mu, sd, 3, 0) is True) == bool(g["ucb_stop"])
    ia, ib = T.overlap(g["ov_a"], g["ov_b"])
    assert np.array_equal(ia, g["ov_ia"]) and np.array_equal(ib, g["ov_ib"])


def test_co2_host_functions_vs_reference_source():
    """init_hyperms, UBC and EI of CO2_example.py (:206-258, :296-306) against vectors produced by
    executing the reference's own function source."""
    from conftest import golden
    from gaussian_process_amd import CO2_example as C2
    g = golden("kernels_bo_co2")
    assert np.array_equal(C2.init_hyperms(5, 11), g["co2_hp"])
    mu, sd = g["co2_bo_mu"], g["co2_bo_sd"]
    assert np.array_equal(np.asarray(C2.UBC(g["co2_hp"], g["co2_hq"], mu, sd)), g["co2_ubc"])
    assert np.array_equal(np.asarray(C2.EI(g["co2_hq"], mu, sd, g["co2_hp_lml"])), g["co2_ei"])
    # candidate sampling: 11 columns inside the reference's [0.3, 1.5] x book window, none already tried
    import random
    random.seed(1)
    cand = C2.random_sample_test_parms(20, g["co2_hp"])
    assert cand.shape == (20, 11)
    assert np.all(cand >= 0.3 * C2.HYPERMS_BOOK - 1e-12) and np.all(cand <= 1.5 * C2.HYPERMS_BOOK + 1e-12)


# ---- host-only pieces of the BO loop against vectors produced by EXECUTING the reference's
# own Python-2 source (lib2to3 print/repr fixers, oracle/make_golden.py bo_loop_cases)
def test_PI_and_candidate_sampling_vs_reference_source():
    import random
    from conftest import golden
    from gaussian_process_amd import tune_hyperparms_regression as T
    g = golden("kernels_bo_loops")
    random.seed(7)
    got = T.PI(g["pi_params"], g["pi_mu"], g["pi_sd"], g["pi_done"], g["pi_y"], 3, 0)
    assert np.array_equal(np.ravel(got), np.ravel(g["pi_next"]))
    random.seed(8)                                   # three candidates tie at cdf == 1: random.randint picks
    got = T.PI(g["pi_params"], g["pi_mu_tie"], g["pi_sd"], g["pi_done"], g["pi_y"], 3, 0)
    assert np.array_equal(np.ravel(got), np.ravel(g["pi_next_tie"]))
    assert T.PI(g["pi_params"], g["pi_mu"] - 1e4, g["pi_sd"], g["pi_done"], g["pi_y"], 3, 0) is True and bool(g["pi_stop"])
    random.seed(11)
    out = T.random_gen_test_parms(100, g["rg_done"])
    assert out.shape == (100, 1) and np.array_equal(out, g["rg_out"])


def test_bo_loop_decisions_vs_reference_source_given_its_surrogate():
    """Every iteration of the reference's tune_hyperparms_second run: fed the reference's own surrogate mean / sd
    and candidate set, the drop-in's acquisition picks the reference's next lengthscale (host logic only)."""
    import random
    from conftest import golden
    from gaussian_process_amd import tune_hyperparms_regression as T
    g = golden("kernels_bo_loops")
    random.seed(5)
    for k in range(int(g["lp_iters"])):
        cand = T.random_gen_test_parms(100, g["lp%d_l" % k])
        assert np.array_equal(cand, g["lp%d_cand" % k])          # same random.sample stream as the reference
        nxt = T.PI(cand, g["lp%d_mu" % k], g["lp%d_sd" % k], g["lp%d_l" % k], g["lp%d_lml" % k], 3, k)
        want = float(g["lp%d_next" % k])
        assert (nxt is True and want < 0) or float(np.ravel(nxt)[0]) == want
