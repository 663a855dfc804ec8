"""Pin the CPU oracle (oracle/gp_oracle.py, rbf_oracle.c) against the golden vectors
produced by the reference itself (oracle/make_golden.py).  Runs without a GPU.

Tolerances: the oracle issues the same NumPy/LAPACK calls as the reference, so on
the machine that generated the goldens it is bit-identical; the slack below only
covers a different host (SIMD width of np.exp, OpenBLAS thread count)."""
import numpy as np
import pytest

from conftest import golden, golden_names

K_RTOL = 1e-15          # exp() may differ by < 1 ulp between hosts
MU_ATOL = 1e-9
SD_ATOL = 1e-9
LML_RTOL = 1e-11
FPOST_ATOL = 1e-6       # Cholesky of a jitter-regularised, near-singular posterior covariance


@pytest.mark.parametrize("name", golden_names())
def test_rbf_kernel_matches_reference(oracle, name):
    g = golden(name)
    K = oracle.RBF_kernel(g["X"], g["X"], float(g["sigma"]), float(g["ell"]))
    assert np.allclose(K[:16, :16], g["K_corner"], rtol=K_RTOL, atol=0)
    assert np.allclose(K[-1], g["K_lastrow"], rtol=K_RTOL, atol=0)
    assert np.allclose(K.sum(1), g["K_rowsum"], rtol=1e-14, atol=0)
    assert abs(np.linalg.norm(K) - g["K_fro"]) <= 1e-14 * g["K_fro"]
    assert np.array_equal(np.diag(K), np.full(len(K), float(g["sigma"]) ** 2))   # exact diagonal
    assert np.array_equal(K, K.T)                                                 # exact symmetry
    # the chunked and the C restatements carry the same arithmetic
    assert np.allclose(oracle.RBF_kernel_chunked(g["X"], g["X"], float(g["sigma"]), float(g["ell"]), rows=37),
                       K, rtol=K_RTOL, atol=0)
    assert np.allclose(oracle.RBF_kernel_c(g["X"], g["X"], float(g["sigma"]), float(g["ell"])),
                       K, rtol=4e-16, atol=0)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cfg1_prediction_reproduces_reference(oracle, seed):
    """BASELINE config 1: GP_regression.py on N=512, d=1 sine data (NumPy path)."""
    g = golden("cfg1_seed%d" % seed)
    np.random.seed(seed)
    f, X, y, Xs = oracle.dataset_generator(512, 100)
    assert np.array_equal(X, g["X"]) and np.array_equal(y, g["y"]) and np.array_equal(Xs, g["Xs"])
    mu, sd, fpost = oracle.prediction(X, Xs, y, 'rbf', 1, 10)
    assert np.allclose(mu, g["mu"], rtol=0, atol=MU_ATOL)
    assert np.allclose(sd, g["sd"], rtol=0, atol=SD_ATOL)
    assert np.allclose(fpost, g["f_post"], rtol=0, atol=FPOST_ATOL)
    lml = oracle.compute_mar_likelihood(X, Xs, y, 1, 1)
    assert abs(lml - g["lml"]) <= LML_RTOL * abs(g["lml"])


@pytest.mark.parametrize("name", golden_names("d"))
def test_multid_posterior_and_lml(oracle, name):
    g = golden(name)
    X, y, Xs, ell = g["X"], g["y"], g["Xs"], float(g["ell"])
    np.random.seed(int(g["seed"]))
    mu, sd, fpost = oracle.prediction(X, Xs, y, 'rbf', ell, 3)
    assert np.allclose(mu, g["mu"], rtol=0, atol=MU_ATOL)
    assert np.allclose(sd, g["sd"], rtol=0, atol=SD_ATOL)
    assert np.allclose(fpost, g["f_post"], rtol=0, atol=FPOST_ATOL)
    assert abs(oracle.compute_mar_likelihood(X, Xs, y, 1, ell) - g["lml"]) <= LML_RTOL * abs(g["lml"])
    assert abs(oracle.compute_mar_likelihood(X, Xs, y, float(g["sigma2"]), float(g["ell2"])) - g["lml2"]) \
        <= LML_RTOL * abs(g["lml2"])
    p = oracle.posterior(X, Xs, y, 1, ell, 0.0005)
    amax = np.abs(g["alpha"]).max()
    assert np.allclose(p["alpha"], g["alpha"], rtol=0, atol=1e-9 * amax)
    assert np.allclose(np.diagonal(p["L"]), g["diagL"], rtol=1e-12, atol=0)
    # the memory-feasible restatement (true triangular solves) agrees with the LU-based reference path
    fz = oracle.fit_predict_feasible(X, Xs, y, 1, ell, 0.0005)
    assert np.allclose(fz["mu"], g["mu"], rtol=0, atol=1e-9)
    assert np.allclose(np.sqrt(fz["var"]), g["sd"], rtol=0, atol=1e-9)
    assert abs(fz["lml"] - g["lml"]) <= 1e-10 * abs(g["lml"])


def test_edge_cases(oracle):
    e = golden("edge_cases")
    # N = 1, n = 1
    np.random.seed(5)
    mu, sd, fp = oracle.prediction(e["n1_X"], e["n1_Xs"], e["n1_y"], 'rbf', 1, 2)
    assert np.allclose(mu, e["n1_mu"], atol=1e-14) and np.allclose(sd, e["n1_sd"], atol=1e-14)
    assert np.allclose(fp, e["n1_fpost"], atol=1e-12)
    assert abs(oracle.compute_mar_likelihood(e["n1_X"], None, e["n1_y"], 1, 1) - e["n1_lml"]) < 1e-13
    # duplicate rows (exactly singular K rescued by + s I), ragged sizes
    for tag, seed, nf in (("dup", 6, 1), ("rag", 7, 2)):
        np.random.seed(seed)
        mu, sd, fp = oracle.prediction(e[tag + "_X"], e[tag + "_Xs"], e[tag + "_y"], 'rbf', float(e[tag + "_ell"]), nf)
        assert np.allclose(mu, e[tag + "_mu"], atol=1e-9)
        assert np.allclose(sd, e[tag + "_sd"], atol=1e-9, equal_nan=True)
        assert np.allclose(fp, e[tag + "_fpost"], atol=FPOST_ATOL)
        lml = oracle.compute_mar_likelihood(e[tag + "_X"], None, e[tag + "_y"], 1, float(e[tag + "_ell"]))
        assert abs(lml - e[tag + "_lml"]) <= 1e-10 * abs(e[tag + "_lml"])
    # rectangular kernel with an array-valued lengthscale (l[i], tune_hyperparms_regression.py:369)
    K = oracle.RBF_kernel(e["rbf_A"], e["rbf_B"], float(e["rbf_sigma"]), np.array([float(e["rbf_ell"])]))
    assert K.shape == (33, 65) and np.allclose(K, e["rbf_K"], rtol=K_RTOL, atol=0)
    # not positive definite -> LinAlgError, as np.linalg.cholesky does in the reference
    assert int(e["npd_raised"]) == 1
    Kn = oracle.RBF_kernel(e["npd_X"], e["npd_X"], 1, float(e["npd_ell"]))
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.cholesky(Kn + float(e["npd_shift"]) * np.eye(len(Kn)))


def test_linear_and_periodic_kernels(oracle):
    """lin_kernel / per_kernel and prediction() through them."""
    g = golden("kernels_lin_per")
    X, Xs = g["X"], g["Xs"]
    assert np.allclose(oracle.lin_kernel(X[:40], Xs, float(g["c"])), g["K_lin"], rtol=1e-15, atol=1e-15)
    assert np.allclose(oracle.per_kernel(X[:40], Xs, (float(g["p"]), float(g["l"]))), g["K_per"], rtol=1e-15, atol=0)
    np.random.seed(41)
    mu, sd, fp = oracle.prediction_other(X, Xs, g["y_lin"], 'lin', float(g["c"]), 2)
    assert np.allclose(mu, g["lin_mu"], atol=1e-9) and np.allclose(sd, g["lin_sd"], atol=1e-9)
    assert np.allclose(fp, g["lin_fpost"], atol=FPOST_ATOL)
    np.random.seed(42)
    mu, sd, fp = oracle.prediction_other(X, Xs, g["y_per"], 'per', (float(g["p"]), float(g["l"])), 2)
    assert np.allclose(mu, g["per_mu"], atol=1e-9) and np.allclose(sd, g["per_sd"], atol=1e-9)
    assert np.allclose(fp, g["per_fpost"], atol=FPOST_ATOL)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gradient_ascent_terms(oracle, tag):
    """LML, the two gradient traces and one ascent step against the
    vectors produced with the imported RBF_kernel and the reference's statements
    (tune_hyperparms_regression.py:123-145, :43-63)."""
    g = golden("kernels_grad")
    X, y = g[tag + "_X"], g[tag + "_y"]
    sigma, l = float(g[tag + "_sigma"]), float(g[tag + "_l"])
    lml, l_var, sigma_var, alpha, K_y = oracle.lml_and_gradient(X, y, sigma, l)
    assert abs(lml - float(g[tag + "_lml"])) <= 1e-12 * abs(float(g[tag + "_lml"]))
    assert np.allclose(alpha, g[tag + "_alpha"], rtol=0, atol=1e-9 * np.abs(alpha).max())
    # the traces cancel two terms of size ~|tr(K_y^-1 dK)|; same LAPACK calls, so nearly bitwise
    assert abs(l_var - float(g[tag + "_l_var"])) <= 1e-9 * max(1.0, abs(float(g[tag + "_l_var"])))
    assert abs(sigma_var - float(g[tag + "_sigma_var"])) <= 1e-9 * max(1.0, abs(float(g[tag + "_sigma_var"])))
    _, l_next = oracle.gradient_ascent(X, X, sigma, l, alpha.reshape(-1, 1), K_y)
    assert abs(l_next - float(g[tag + "_l_next"])) <= 1e-11 * max(1.0, abs(l_next))
    # finite-difference check of the restated formulas themselves
    h = 1e-5
    fd_l = (oracle.compute_mar_likelihood(X, None, y, sigma, l + h) -
            oracle.compute_mar_likelihood(X, None, y, sigma, l - h)) / (2 * h)
    fd_s = (oracle.compute_mar_likelihood(X, None, y, sigma + h, l) -
            oracle.compute_mar_likelihood(X, None, y, sigma - h, l)) / (2 * h)
    assert abs(fd_l - l_var) <= 1e-5 * max(1.0, abs(l_var))
    assert abs(fd_s - sigma_var) <= 1e-5 * max(1.0, abs(sigma_var))


def test_gradient_ascent_loop_converges(oracle):
    """tune_hyperparms_first (:104-162) on the reference's own data shape: the loop stops on
    |dLML| <= 1e-3 and lands on a stationary lengthscale."""
    np.random.seed(3)
    f, X, y, Xs = oracle.dataset_generator(40, 25)
    mu, sd, fp, lml, l, it = oracle.tune_hyperparms_first(X, Xs, y, 2, 1, np.array([1.5]), max_iter=400)
    assert it < 400 and mu.shape == (25,) and sd.shape == (25,) and fp.shape == (25, 2)
    _, l_var, _, _, _ = oracle.lml_and_gradient(X, y, 1, float(l[0]))
    assert abs(0.01 * l_var) < 1e-2          # the step that would follow is small


def test_bayesian_opt_vs_reference_function(oracle):
    """tune_hyperparms_regression.bayesian_opt (:67-101), vectors from the reference's function itself."""
    g = golden("kernels_bo_co2")
    np.random.seed(21)
    mu, sd, fp = oracle.bayesian_opt(g["bo_X"], g["bo_Xs"], g["bo_y"])
    assert np.allclose(mu, g["bo_mu"], rtol=0, atol=1e-9 * np.abs(g["bo_mu"]).max())
    assert np.allclose(sd, g["bo_sd"], rtol=0, atol=1e-9, equal_nan=True)
    assert np.allclose(fp, g["bo_fpost"], rtol=0, atol=1e-6 * np.abs(g["bo_fpost"]).max())


def test_co2_composite_kernel_and_paths(oracle):
    """SURVEY.md section 8f row f4, second half: CO2_example.py's covariance_function, compute_mar_likelihood,
    make_prediction and bayesian_opt; vectors produced by executing the reference's own
    function source (oracle/make_golden.py)."""
    g = golden("kernels_bo_co2")
    th, X, y, Xs = g["co2_theta"], g["co2_X"], g["co2_y"], g["co2_Xs"]
    K = oracle.co2_covariance_function(X, X, th)
    assert np.array_equal(K[:16, :16], g["co2_K_corner"]) and np.array_equal(K[-1], g["co2_K_lastrow"])
    assert np.array_equal(oracle.co2_covariance_function(X, Xs, th), g["co2_Ks"])
    assert np.array_equal(oracle.co2_covariance_function(X[:48], Xs, th), g["co2_Ksq"])     # square: delta added
    assert oracle.co2_compute_mar_likelihood(X, y, th) == float(g["co2_lml"])
    np.random.seed(31)
    mu, sd, fp = oracle.co2_make_prediction(X, Xs, y, th)
    assert np.array_equal(mu, g["co2_mu"]) and np.array_equal(sd, g["co2_sd"], equal_nan=True)
    assert np.array_equal(fp, g["co2_fpost"])
    mu, sd = oracle.co2_bayesian_opt(g["co2_hp"], g["co2_hq"], g["co2_hp_lml"])
    assert np.array_equal(mu, g["co2_bo_mu"]) and np.array_equal(sd, g["co2_bo_sd"], equal_nan=True)
    assert np.array_equal(oracle.co2_covariance_function(g["co2_hp"], g["co2_hq"], g["co2_hp"][0]), g["co2_Khp"])


def test_blocked_oracle_equals_the_lapack_one():
    """oracle.fit_predict_blocked (the factorisation written out in 4096-wide blocks, used once for the N = 131072 fixture
    of BASELINE config 4) against oracle.fit_predict_feasible (LAPACK dpotrf on the whole matrix): two orders of the same
    sums, so they agree to cond x eps; ragged last block"""
    import gp_oracle as O
    X, y, Xs = O.synthetic_problem(1500, 16, 40, seed=5)
    a = O.fit_predict_feasible(X, Xs, y, 1.0, 2.8, 5e-4)
    b = O.fit_predict_blocked(X, Xs, y, 1.0, 2.8, 5e-4, block=384)
    assert np.max(np.abs(a["mu"] - b["mu"])) <= 1e-10 and np.max(np.abs(a["var"] - b["var"])) <= 1e-11
    assert abs(a["lml"] - b["lml"]) <= 1e-11 * abs(a["lml"])
    assert np.max(np.abs(a["alpha"] - b["alpha"])) <= 1e-9 * np.max(np.abs(a["alpha"]))
    assert np.max(np.abs(a["diagL"] - b["diagL"]) / a["diagL"]) <= 1e-12
