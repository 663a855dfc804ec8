"""Generate tests/golden/*.npz from the reference itself.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (needs /root/reference); the GPU box never
sees the reference, it gets the committed .npz fixtures.  The fixtures hold
data only -- inputs and the reference's outputs on them.

What is imported:  /root/reference/GP_regression.py  (Python 3 clean).
What cannot be:    tune_hyperparms_regression.py and CO2_example.py are Python-2 modules
                   (print statements; first SyntaxError at tune...:150), so they cannot be
                   imported whole.  Most of their FUNCTIONS are valid Python 3 on their own,
                   though: `ref_functions` below reads the file, cuts out the named top-level
                   `def` blocks and executes exactly that source text (nothing is copied into
                   this repository) with NumPy and the imported RBF_kernel in scope.  That is
                   how compute_mar_likelihood, gradient_ascent, bayesian_opt and the CO2
                   example's covariance_function / compute_mar_likelihood / bayesian_opt /
                   make_prediction produce the vectors here.  Intermediate quantities the
                   reference does not return (diag L, m, alpha, the two gradient traces) are
                   produced by issuing the reference's statements with the same NumPy calls,
                   and asserted against the executed functions where they overlap.

usage:  MPLBACKEND=Agg python oracle/make_golden.py
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import GP_regression as REF  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def ref_functions(path, names, namespace):
    """Execute the source of the named top-level functions of a reference file (read as text,
    in this container only) inside `namespace`; returns the namespace."""
    import re
    lines = open(path, newline=None).read().split("\n")
    blocks, cur, buf = {}, None, []
    for ln in lines:
        m = re.match(r"^def\s+(\w+)\s*\(", ln)
        if m:
            if cur:
                blocks[cur] = "\n".join(buf)
            cur, buf = m.group(1), [ln]
        elif cur:
            if re.match(r"^\S", ln):
                blocks[cur] = "\n".join(buf)
                cur, buf = None, []
            else:
                buf.append(ln)
    if cur:
        blocks[cur] = "\n".join(buf)
    for name in names:
        exec(compile(blocks[name], "%s:%s" % (path, name), "exec"), namespace)
    return namespace


TUNE = ref_functions("/root/reference/tune_hyperparms_regression.py",
                     ["compute_mar_likelihood", "gradient_ascent", "bayesian_opt"],
                     {"np": np, "RBF_kernel": REF.RBF_kernel})


def ref_lml(X, y, sigma, l, s=0.0005):
    # tune_hyperparms_regression.py:303-312 issued with the imported RBF_kernel
    n = len(X)
    K = REF.RBF_kernel(X, X, sigma, l)
    L = np.linalg.cholesky(K + s * np.eye(n))
    m = np.linalg.solve(L, y)
    alpha = np.linalg.solve(L.T, m)
    lml = -.5 * np.dot(y.T, alpha) - np.log(np.diagonal(L)).sum(0) - n / 2.0 * np.log(2 * np.pi)
    if s == 0.0005:
        # the reference's own function (executed from its source) returns the same number
        assert TUNE["compute_mar_likelihood"](X, None, y, sigma, l) == lml
    return lml, L, m, alpha


def k_summary(K):
    return dict(K_corner=K[:16, :16].copy(), K_rowsum=K.sum(1), K_fro=np.linalg.norm(K),
                K_lastrow=K[-1].copy())


def cfg1(seed):
    """BASELINE config 1: GP_regression.py defaults at N=512, n=100, d=1, l=1."""
    np.random.seed(seed)
    f, X, y, Xs = REF.dataset_generator(512, 100)
    state_after_data = np.random.get_state()
    mu, sd, fpost = REF.prediction(X, Xs, y, 'rbf', 1, 10)
    np.random.set_state(state_after_data)
    normals = np.random.normal(size=(100, 10))      # the draw prediction() made (:155)
    K = REF.RBF_kernel(X, X, 1, 1)
    lml, L, m, alpha = ref_lml(X, y, 1, 1)
    np.savez_compressed(os.path.join(OUT, "cfg1_seed%d.npz" % seed), seed=seed, X=X, y=y, Xs=Xs,
                        mu=mu, sd=sd, f_post=fpost, normals=normals, lml=lml, diagL=np.diagonal(L),
                        m=m, alpha=alpha, ell=1.0, sigma=1.0, s=0.0005, **k_summary(K))


def dcase(name, N, d, n, lo, hi, ell, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(lo, hi, (N, d))
    y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
    Xs = rng.uniform(lo, hi, (n, d))
    np.random.seed(seed)
    mu, sd, fpost = REF.prediction(X, Xs, y, 'rbf', ell, 3)
    np.random.seed(seed)
    normals = np.random.normal(size=(n, 3))
    K = REF.RBF_kernel(X, X, 1, ell)
    Ks = REF.RBF_kernel(X, Xs, 1, ell)
    lml, L, m, alpha = ref_lml(X, y, 1, ell)
    # a second hyper-parameter triple through the LML formula (sigma != 1)
    lml2, _, _, _ = ref_lml(X, y, 1.5, 0.75 * ell)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, X=X, y=y, Xs=Xs, mu=mu, sd=sd,
                        f_post=fpost, normals=normals, lml=lml, lml2=lml2, sigma2=1.5, ell2=0.75 * ell,
                        diagL=np.diagonal(L), m=m, alpha=alpha, ell=float(ell), sigma=1.0, s=0.0005,
                        Ks_corner=Ks[:16, :16].copy(), Ks_colsum=Ks.sum(0), **k_summary(K))


def edge_cases():
    out = {}
    # N=1, n=1
    X = np.array([[0.3]]); y = np.array([0.7]); Xs = np.array([[0.1]])
    np.random.seed(5)
    mu, sd, fp = REF.prediction(X, Xs, y, 'rbf', 1, 2)
    out.update(n1_X=X, n1_y=y, n1_Xs=Xs, n1_mu=mu, n1_sd=sd, n1_fpost=fp,
               n1_lml=ref_lml(X, y, 1, 1)[0])
    # duplicate rows: K exactly singular, rescued by + s I
    rng = np.random.default_rng(11)
    Xd = rng.uniform(-2, 2, (24, 3)); Xd[7] = Xd[3]; Xd[20] = Xd[3]
    yd = np.sin(Xd.sum(1)); Xsd = rng.uniform(-2, 2, (9, 3))
    np.random.seed(6)
    mu, sd, fp = REF.prediction(Xd, Xsd, yd, 'rbf', 1.5, 1)
    out.update(dup_X=Xd, dup_y=yd, dup_Xs=Xsd, dup_mu=mu, dup_sd=sd, dup_fpost=fp,
               dup_lml=ref_lml(Xd, yd, 1, 1.5)[0], dup_ell=1.5)
    # ragged (non tile-multiple) sizes, n > N
    Xr = rng.uniform(-1, 1, (77, 5)); yr = np.cos(Xr[:, 0]) + 0.1 * Xr[:, 1]; Xsr = rng.uniform(-1, 1, (131, 5))
    np.random.seed(7)
    mu, sd, fp = REF.prediction(Xr, Xsr, yr, 'rbf', 0.9, 2)
    out.update(rag_X=Xr, rag_y=yr, rag_Xs=Xsr, rag_mu=mu, rag_sd=sd, rag_fpost=fp,
               rag_lml=ref_lml(Xr, yr, 1, 0.9)[0], rag_ell=0.9)
    # RBF_kernel alone on rectangular / scalar-as-array hyper-parameters (l[i] at tune...:369)
    A = rng.normal(size=(33, 4)); B = rng.normal(size=(65, 4))
    out.update(rbf_A=A, rbf_B=B, rbf_K=REF.RBF_kernel(A, B, 1.7, np.array([0.6]))[:, :],
               rbf_sigma=1.7, rbf_ell=0.6)
    # non-PD: negative "noise" big enough to break Cholesky -> LinAlgError in the reference
    Xn = rng.uniform(-1, 1, (40, 2)); Kn = REF.RBF_kernel(Xn, Xn, 1, 2.0)
    try:
        np.linalg.cholesky(Kn - 0.5 * np.eye(40))
        raised = 0
    except np.linalg.LinAlgError:
        raised = 1
    out.update(npd_X=Xn, npd_ell=2.0, npd_shift=-0.5, npd_raised=raised)
    np.savez_compressed(os.path.join(OUT, "edge_cases.npz"), **out)


def other_kernels():
    """the reference's lin_kernel / per_kernel and prediction() through them
    (1-D inputs, as the reference uses them)."""
    rng = np.random.default_rng(31)
    X = rng.uniform(-5, 5, (200, 1)); Xs = np.linspace(-5, 5, 37).reshape(-1, 1)
    y_lin = 0.7 * X[:, 0] - 0.3 + np.sqrt(5e-4) * rng.standard_normal(200)
    y_per = np.sin(2 * np.pi * X[:, 0] / 3.0) + np.sqrt(5e-4) * rng.standard_normal(200)
    out = dict(X=X, Xs=Xs, y_lin=y_lin, y_per=y_per, c=0.5, p=3.0, l=1.2)
    out["K_lin"] = REF.lin_kernel(X[:40], Xs, 0.5)
    out["K_per"] = REF.per_kernel(X[:40], Xs, (3.0, 1.2))
    np.random.seed(41)
    out["lin_mu"], out["lin_sd"], out["lin_fpost"] = REF.prediction(X, Xs, y_lin, 'lin', 0.5, 2)
    np.random.seed(42)
    out["per_mu"], out["per_sd"], out["per_fpost"] = REF.prediction(X, Xs, y_per, 'per', (3.0, 1.2), 2)
    np.savez_compressed(os.path.join(OUT, "kernels_lin_per.npz"), **out)


def grad_cases():
    """the gradient-ascent terms.  tune_hyperparms_regression.py is Python-2
    syntax (not importable), so its lines :123-129, :141, :144 and :43-57 are issued here with
    the IMPORTED RBF_kernel and the same NumPy calls (the sigma term is the commented-out
    code at :46-51)."""
    out = {}
    for tag, N, d, lo, hi, sigma, l in (("a", 96, 1, -5.0, 5.0, 1.0, 1.0), ("b", 200, 3, -2.0, 2.0, 1.3, 0.8),
                                        ("c", 384, 8, -1.0, 1.0, 1.0, 2.0)):
        rng = np.random.default_rng(500 + N)
        X = rng.uniform(lo, hi, (N, d))
        y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N)
        s = 0.0005
        K_train = REF.RBF_kernel(X, X, sigma, l)                                    # :123
        L = np.linalg.cholesky(K_train + s * np.eye(N))                             # :127
        m = np.linalg.solve(L, y)                                                   # :128
        alpha = np.linalg.solve(L.T, m)                                             # :129
        lml = -.5 * np.dot(y.T, alpha) - np.log(np.diagonal(L)).sum(0) - N / 2.0 * np.log(2 * np.pi)   # :141
        K_y = np.dot(np.linalg.inv(L.T), np.linalg.inv(L))                          # :144
        al = alpha.reshape(-1, 1)
        a = b = X
        sqdist = ((a[:, :, None] - b[:, :, None].T) ** 2).sum(1)                    # :43
        sigma_grad = 2 * sigma * np.exp(-.5 * sqdist / (l ** 2))                    # :48
        sigma_var = .5 * np.diagonal(np.dot(np.dot(al, al.T) - K_y, sigma_grad)).sum()   # :49-51
        l_grad = sigma ** 2 * np.exp(-.5 * sqdist / (l ** 2)) * (sqdist / l ** 3)   # :54
        l_var = .5 * np.diagonal(np.dot(np.dot(al, al.T) - K_y, l_grad)).sum()      # :55-57
        s_ref, l_ref = TUNE["gradient_ascent"](X, X, sigma, l, al, K_y)     # the reference's function itself (:31-64)
        assert s_ref == sigma and l_ref == l + 0.01 * l_var
        out.update({tag + "_X": X, tag + "_y": y, tag + "_sigma": sigma, tag + "_l": l, tag + "_lml": lml,
                    tag + "_alpha": alpha, tag + "_l_var": l_var, tag + "_sigma_var": sigma_var,
                    tag + "_l_next": l + 0.01 * l_var})                              # :42, :63
        if N <= 200:
            out[tag + "_Kyinv"] = K_y          # the larger inverse is recomputed by the tests (fixture size)
    np.savez_compressed(os.path.join(OUT, "kernels_grad.npz"), **out)


def bo_and_co2_cases():
    """(1) tune_hyperparms_regression.bayesian_opt (:67-101) executed from the reference source.
    (2) CO2_example.py: covariance_function (+ kernel_1..4), compute_mar_likelihood, bayesian_opt and
    make_prediction executed from the reference source (second half).  The Mauna Loa data
    set is a network fetch in the reference (:405) and is not used: inputs are synthetic with
    the same shape (monthly decimal years, 1-D)."""
    out = {}
    rng = np.random.default_rng(77)
    # (1)
    lt = np.array([0.4, 1.1, 2.3, 3.1, 4.6]).reshape(-1, 1)
    lq = np.sort(rng.uniform(0.02, 5, 60)).reshape(-1, 1)
    yl = np.array([-310.2, 210.5, 402.75, 380.1, 150.9])
    np.random.seed(21)
    mu, sd, fp = TUNE["bayesian_opt"](lt, lq, yl)
    out.update(bo_X=lt, bo_Xs=lq, bo_y=yl, bo_mu=mu, bo_sd=sd, bo_fpost=fp)
    # (2)
    CO2 = ref_functions("/root/reference/CO2_example.py",
                        ["kernel_1", "kernel_2", "kernel_3", "kernel_4", "covariance_function",
                         "compute_mar_likelihood", "bayesian_opt", "make_prediction", "init_hyperms"],
                        {"np": np})
    book = np.array([66, 67, 2.4, 90, 1.3, .66, 1.2, .78, .18, 1.6, .19])             # :120, :323
    X = (1958.0 + np.arange(360) / 12.0 + 0.04).reshape(-1, 1)
    y = 0.11 * (X[:, 0] - 1958) ** 1.5 + 3 * np.sin(2 * np.pi * X[:, 0]) + 0.3 * rng.standard_normal(360)
    y = y - np.mean(y)
    Xs = (1988.0 + np.arange(48) / 12.0 + 0.04).reshape(-1, 1)
    K = CO2["covariance_function"](X, X, book)
    Ks = CO2["covariance_function"](X, Xs, book)
    Ksq = CO2["covariance_function"](X[:48], Xs, book)          # square but a != b: the delta quirk (:58)
    out.update(co2_theta=book, co2_X=X, co2_y=y, co2_Xs=Xs, co2_K_corner=K[:16, :16].copy(), co2_K_rowsum=K.sum(1),
               co2_K_lastrow=K[-1].copy(), co2_Ks=Ks, co2_Ksq=Ksq,
               co2_lml=CO2["compute_mar_likelihood"](X, y, book))
    np.random.seed(31)
    mu, sd, fp = CO2["make_prediction"](X, Xs, y, book)
    out.update(co2_mu=mu, co2_sd=sd, co2_fpost=fp)
    hp = CO2["init_hyperms"](5, 11)                                                    # :296-306
    hq = hp[0] + rng.uniform(-3, 6, (40, 11))
    lmls = np.array([CO2["compute_mar_likelihood"](X, y, h) for h in hp])              # :338-339
    mu, sd = CO2["bayesian_opt"](hp, hq, lmls)                                         # :341 (multi-D branch :83)
    out.update(co2_hp=hp, co2_hq=hq, co2_hp_lml=lmls, co2_bo_mu=mu, co2_bo_sd=sd,
               co2_Khp=CO2["covariance_function"](hp, hq, hp[0]))
    # the host-side acquisition functions of the CO2 loop (:206-258), executed from source
    from scipy.stats import norm
    ACQ = ref_functions("/root/reference/CO2_example.py", ["UBC", "EI"], {"np": np, "norm": norm})
    out["co2_ubc"] = np.asarray(ACQ["UBC"](hp, hq, mu, sd))
    out["co2_ei"] = np.asarray(ACQ["EI"](hq, mu, sd, lmls))
    np.savez_compressed(os.path.join(OUT, "kernels_bo_co2.npz"), **out)


def acquisition_cases():
    """UCB / EI / TS (tune_hyperparms_regression.py:207-268) and overlap (:316-328)
    executed from the reference source (PI has a Python-2 print and cannot be)."""
    from scipy.stats import norm
    import matplotlib.pyplot as plt
    A = ref_functions("/root/reference/tune_hyperparms_regression.py", ["UCB", "EI", "TS", "overlap"],
                      {"np": np, "norm": norm, "plt": plt, "prediction": REF.prediction})
    rng = np.random.default_rng(91)
    done = np.array([0.4, 1.1, 2.3, 3.1, 4.6])
    y = np.array([-310.2, 210.5, 402.75, 380.1, 150.9])
    params = np.sort(rng.uniform(0.02, 5, 60)).reshape(-1, 1)
    np.random.seed(21)
    mu, sd, _ = TUNE["bayesian_opt"](done.reshape(-1, 1), params, y)
    out = dict(done=done, y=y, params=params, mu=mu, sd=sd)
    out["ucb"] = np.asarray(A["UCB"](done, params, mu, sd, 3, 0))
    out["ei"] = np.asarray(A["EI"](params, mu, sd, done, y, 3, 0))
    np.random.seed(22)
    out["ts"] = np.asarray(A["TS"](done, params, y, 3, 0))
    # UCB's stop rule: proposing the last evaluated point again returns True
    params2 = params.copy(); j = int(np.argmax(mu + 0.001 * sd)); done2 = np.append(done, params2[j, 0])
    out["ucb_stop"] = np.asarray(A["UCB"](done2, params2, mu, sd, 3, 0) is True)
    grid = np.linspace(0.01, 5, 75)
    a = np.array([grid[3], 0.777, grid[40], grid[74]])
    ia, ib = A["overlap"](a, grid)
    out.update(ov_a=a, ov_b=grid, ov_ia=ia, ov_ib=ib)
    np.savez_compressed(os.path.join(OUT, "kernels_acq.npz"), **out)


def py2_functions(path, names, namespace):
    """ref_functions for def blocks that contain Python-2 print statements / backtick repr: lib2to3's print and
    repr fixers run over the block's text (in memory, in this container only), then the text is executed."""
    import re
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3.refactor import RefactoringTool
        tool = RefactoringTool(["lib2to3.fixes.fix_print", "lib2to3.fixes.fix_repr"])
    text = open(path, newline=None).read()
    for name in names:
        m = re.search(r"^def %s\(.*?(?=^\S|\Z)" % re.escape(name), text, re.S | re.M)
        src = str(tool.refactor_string(m.group(0).rstrip("\n") + "\n", name))
        exec(compile(src, "%s:%s" % (path, name), "exec"), namespace)
    return namespace


class Py3Random:
    """`random` as the reference's functions see it: Python 3's random.sample refuses an ndarray population
    (tune_hyperparms_regression.py:343, CO2_example.py:121), so sample() lists it first -- the drop-ins make the
    same call.  Everything else is the module's own."""
    def __init__(self):
        import random
        self._r = random

    def sample(self, population, k):
        return self._r.sample(list(population), k)

    def __getattr__(self, name):
        return getattr(self._r, name)


class NpCompat:
    """`np` as the reference's 2017-era code expects it: np.delete accepted the float-typed EMPTY index array
    that overlap() returns when nothing overlaps (tune_hyperparms_regression.py:328, 342); NumPy 2 raises
    IndexError for it.  Only that call is adapted (index array cast to int, what the drop-ins do); every other
    attribute is NumPy's own."""
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def delete(arr, obj, axis=None):
        obj = np.asarray(obj)
        if obj.dtype.kind == "f":
            obj = obj.astype(int)
        return np.delete(arr, obj, axis)


class Recorder:
    """stand-in for print inside executed reference code: keeps what was printed"""
    def __init__(self):
        self.items = []

    def __call__(self, *args):
        self.items.append(args[0] if len(args) == 1 else args)


def bo_loop_cases():
    """SURVEY.md section 8f row f3, the parts with Python-2 prints: PI (:165-204), random_gen_test_parms (:331-346),
    tune_hyperparms_second (:349-395) of tune_hyperparms_regression.py and tune_hyperparameters_BO
    (CO2_example.py:309-371), executed from the reference's own source through lib2to3 (print / repr fixers
    only) with seeded `random` and `np.random`.  Plot calls go to a throw-away Agg figure, plot_BO is a no-op."""
    import random
    from scipy.stats import norm
    import matplotlib.pyplot as plt
    rnd = Py3Random()
    rec = Recorder()
    npc = NpCompat()
    ns = {"np": npc, "norm": norm, "plt": plt, "random": rnd, "RBF_kernel": REF.RBF_kernel,
          "prediction": REF.prediction, "print": rec, "plot_BO": lambda *a, **k: None}
    path = "/root/reference/tune_hyperparms_regression.py"
    ref_functions(path, ["compute_mar_likelihood", "bayesian_opt", "UCB", "EI", "TS", "overlap", "acquisition_fun"], ns)
    py2_functions(path, ["PI", "random_gen_test_parms", "tune_hyperparms_second"], ns)
    out = {}
    # ---- PI alone (three situations: a unique maximiser, ties broken by random.randint, the early stop)
    rng = np.random.default_rng(191)
    done = np.array([0.4, 1.1, 2.3, 3.1, 4.6])
    y = np.array([-310.2, 210.5, 402.75, 380.1, 150.9])
    params = np.sort(rng.uniform(0.02, 5, 60)).reshape(-1, 1)
    np.random.seed(21)
    mu, sd, _ = ns["bayesian_opt"](done.reshape(-1, 1), params, y)
    random.seed(7)
    out.update(pi_done=done, pi_y=y, pi_params=params, pi_mu=mu, pi_sd=sd,
               pi_next=np.asarray(ns["PI"](params, mu, sd, done, y, 3, 0)))
    mu_t = mu.copy(); mu_t[[5, 17, 40]] = 1e4                       # three candidates with cdf == 1: a tie
    random.seed(8)
    out.update(pi_mu_tie=mu_t, pi_next_tie=np.asarray(ns["PI"](params, mu_t, sd, done, y, 3, 0)))
    out["pi_stop"] = np.asarray(ns["PI"](params, mu - 1e4, sd, done, y, 3, 0) is True)
    # ---- random_gen_test_parms
    random.seed(11)
    out["rg_done"] = np.array([0.5, 3.5, 0.01, 5.0])
    out["rg_out"] = ns["random_gen_test_parms"](100, out["rg_done"])
    # ---- the loop: every surrogate fit, candidate set and chosen point of each iteration
    np.random.seed(3)
    f, X, yv, Xs = REF.dataset_generator(60, 25)
    log = {"bo": [], "cand": [], "next": []}
    bo, gen, acq = ns["bayesian_opt"], ns["random_gen_test_parms"], ns["acquisition_fun"]

    def bo_rec(lt, lq, yl):
        r = bo(lt, lq, yl)
        log["bo"].append((lt.copy(), lq.copy(), yl.copy(), r[0].copy(), r[1].copy()))
        return r

    def gen_rec(n, done_):
        r = gen(n, done_)
        log["cand"].append(r.copy())
        return r

    def acq_rec(*a):
        r = acq(*a)
        log["next"].append(r)
        return r
    ns.update(bayesian_opt=bo_rec, random_gen_test_parms=gen_rec, acquisition_fun=acq_rec)
    l0 = np.array([0.5, 3.5])
    random.seed(5); np.random.seed(5)
    best = ns["tune_hyperparms_second"](X, Xs, yv, 1, 1, l0.copy())
    out.update(lp_X=X, lp_y=yv, lp_Xs=Xs, lp_l0=l0, lp_best=best, lp_iters=len(log["next"]))
    for k, (lt, lq, yl, m_, s_) in enumerate(log["bo"]):
        out.update({"lp%d_l" % k: lt.reshape(-1), "lp%d_cand" % k: lq, "lp%d_lml" % k: yl, "lp%d_mu" % k: m_,
                    "lp%d_sd" % k: s_})
    for k, nx in enumerate(log["next"]):
        out["lp%d_next" % k] = np.asarray(-1.0 if nx is True else np.ravel(nx)[0])
    out["lp_printed"] = np.array([str(i) for i in rec.items])
    # ---- CO2_example.tune_hyperparameters_BO
    rec2 = Recorder()
    ns2 = {"np": npc, "norm": norm, "plt": plt, "random": rnd, "print": rec2}
    cpath = "/root/reference/CO2_example.py"
    ref_functions(cpath, ["kernel_1", "kernel_2", "kernel_3", "kernel_4", "covariance_function", "compute_mar_likelihood",
                          "bayesian_opt", "UBC", "TS", "EI", "PI", "acquisition_fun", "init_hyperms"], ns2)
    ns2["overlap"] = ns["overlap"]                 # CO2_example.py:1 imports it from tune_hyperparms_regression
    py2_functions(cpath, ["random_sample_test_parms", "tune_hyperparameters_BO"], ns2)
    rng = np.random.default_rng(78)
    Xc = (1958.0 + np.arange(240) / 12.0 + 0.04).reshape(-1, 1)
    yc = 0.11 * (Xc[:, 0] - 1958) ** 1.5 + 3 * np.sin(2 * np.pi * Xc[:, 0]) + 0.3 * rng.standard_normal(240)
    yc = yc - np.mean(yc)
    clog = {"bo": [], "next": []}
    cbo, cacq = ns2["bayesian_opt"], ns2["acquisition_fun"]

    def cbo_rec(ht, hq, yl):
        r = cbo(ht, hq, yl)
        clog["bo"].append((ht.copy(), hq[:3].copy(), float(hq.sum()), yl.copy(), r[0].copy(), r[1].copy()))
        return r

    def cacq_rec(*a):
        r = cacq(*a)
        clog["next"].append(np.asarray(r).copy())
        return r
    ns2.update(bayesian_opt=cbo_rec, acquisition_fun=cacq_rec)
    random.seed(9); np.random.seed(9)
    plt.figure()
    best_h = ns2["tune_hyperparameters_BO"](Xc, Xc[:5], yc)
    plt.close("all")
    out.update(co_X=Xc, co_y=yc, co_best=best_h, co_steps=len(clog["next"]))
    out["co_next"] = np.array(clog["next"])                                    # (40, 11): 4 passes x 10 iterations
    out["co_ymax"] = np.array([b[3].max() for b in clog["bo"]])                # what the reference prints per iteration
    out["co_cand_head"] = np.array([b[1] for b in clog["bo"]])                 # first 3 candidates of every iteration
    out["co_cand_sum"] = np.array([b[2] for b in clog["bo"]])
    for k in (0, 9, 10, 39):                                                   # surrogate posteriors of four iterations
        out.update({"co%d_train" % k: clog["bo"][k][0], "co%d_lml" % k: clog["bo"][k][3], "co%d_mu" % k: clog["bo"][k][4],
                    "co%d_sd" % k: clog["bo"][k][5]})
    np.savez_compressed(os.path.join(OUT, "kernels_bo_loops.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for sd_ in (0, 1, 2):
        cfg1(sd_)
    for N in (64, 256, 1024):
        dcase("d8_box1_N%d" % N, N, 8, 48, -1.0, 1.0, 2.0, 100 + N)
        dcase("d8_box5_N%d" % N, N, 8, 48, -5.0, 5.0, 4.0, 200 + N)
    dcase("d16_box1_N384", 384, 16, 40, -1.0, 1.0, 2.8, 316)
    edge_cases()
    other_kernels()
    grad_cases()
    bo_and_co2_cases()
    acquisition_cases()
    bo_loop_cases()
    print("wrote", sorted(os.listdir(OUT)))
