"""Developer check: run the HIP path against the golden fixtures and the oracle and
print the error of every compared quantity (used to calibrate test tolerances)."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O  # noqa: E402
from gaussian_process_amd import GPContext  # noqa: E402
from gaussian_process_amd import GP_regression as G  # noqa: E402

ctx = GPContext(0)
print("mfma probe TF/s", ctx.probe_mfma_f64(), " hbm write GB/s", ctx.probe_hbm_write(1 << 30))


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/*.npz"))):
    g = np.load(f)
    if "edge" in f:
        continue
    X, y, Xs = g["X"], g["y"], g["Xs"]
    ell, sigma, s = float(g["ell"]), float(g["sigma"]), float(g["s"])
    K = ctx.rbf(X, X, sigma, ell)
    Kerr = np.max(np.abs(K[:16, :16] - g["K_corner"]) / g["K_corner"])
    lml = ctx.fit(X, y, sigma, ell, s)
    mu, sd = ctx.predict(Xs)
    alpha = ctx.alpha()
    m = ctx.m()
    dg = ctx.diag()
    L_ = ctx.post_chol(1e-6)
    fp = mu.reshape(-1, 1) + L_ @ g["normals"]
    print("%-22s Kcorner_rel %.1e rowsum %.1e | lml rel %.1e | mu %.1e sd %.1e | alpha rel %.1e m rel %.1e diag rel %.1e | fpost %.1e"
          % (os.path.basename(f), Kerr, rel(K.sum(1), g["K_rowsum"]), abs(lml - g["lml"]) / abs(g["lml"]),
             np.abs(mu - g["mu"]).max(), np.nanmax(np.abs(sd - g["sd"])), rel(alpha, g["alpha"]), rel(m, g["m"]),
             rel(dg, g["diagL"]), np.abs(fp - g["f_post"]).max()))

# mid-size vs feasible oracle
for N, d, n in ((2048, 8, 256), (4096, 8, 512)):
    X, y, Xs = O.synthetic_problem(N, d, n)
    t0 = time.time()
    ref = O.fit_predict_feasible(X, Xs, y, 1.0, 2.0, 5e-4)
    t1 = time.time()
    lml = ctx.fit(X, y, 1.0, 2.0, 5e-4)
    mu, var = ctx.predict(Xs, want_sd=False)
    t2 = time.time()
    al = ctx.alpha()
    print("N=%d: cpu %.2fs gpu %.3fs | mu %.1e var %.1e lml rel %.1e alpha rel %.1e" %
          (N, t1 - t0, t2 - t1, np.abs(mu - ref["mu"]).max(), np.abs(var - ref["var"]).max(),
           abs(lml - ref["lml"]) / abs(ref["lml"]), rel(al, ref["alpha"])))
    print("   timers", {k: round(v, 3) for k, v in ctx.timers().items() if v})

# big timing
for N in (8192, 16384, 32768):
    X, y, Xs = O.synthetic_problem(N, 8, 4096)
    ctx.set_train(X, y)
    ctx.set_test(Xs)
    for rep in range(2):
        t0 = time.time()
        lml = ctx.factorize(1.0, 2.0, 5e-4)
        tf = time.time() - t0
        tm = ctx.timers()
        mu, var = ctx.predict_resident(False)
        tp = time.time() - t0 - tf
        tm2 = ctx.timers()
    trail_tf = tm["trail_flops"] / (tm["chol_trail"] * 1e-3) / 1e12 if tm["chol_trail"] else 0
    print("N=%d fit %.3fs predict %.3fs lml %.6f | kbuild %.2fms chol %.1fms (panel %.1f trail %.1f -> %.1f TF/s) | ks %.2f solve_v %.1f meanvar %.2f"
          % (N, tf, tp, lml, tm["kbuild"], tm["chol"], tm["chol_panel"], tm["chol_trail"], trail_tf,
             tm2["ks"], tm2["solve_v"], tm2["meanvar"]), flush=True)
