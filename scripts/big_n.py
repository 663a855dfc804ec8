"""cfg4's problem size on ONE GPU (N=131072, d=16: K + L in place = 137 GB of the 288 GB)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 4096
ell = 2.8 if d == 16 else 2.0
X, y, Xs = O.synthetic_problem(N, d, n)
ctx = GPContext(0)
ctx.set_train(X, y)
t0 = time.perf_counter(); lml = ctx.factorize(1.0, ell, 5e-4); t1 = time.perf_counter()
print("N=%d d=%d fit %.3f s lml %.6f" % (N, d, t1 - t0, lml), flush=True)
# predict at 2048 training inputs + 2048 fresh ones: K_i alpha = y_i - s alpha_i at the training ones
idx = np.random.default_rng(1).choice(N, 2048, replace=False)
Xq = np.vstack([X[idx], Xs[:2048]])
t2 = time.perf_counter(); mu, var = ctx.predict(Xq, want_sd=False); t3 = time.perf_counter()
alpha = ctx.alpha()
res = np.abs(mu[:2048] - (y[idx] - 5e-4 * alpha[idx])).max()
fl = N ** 3 / 3 + N * N * n
print("predict %.3f s; fit+predict %.3f s = %.1f TFLOP/s; max |K_i alpha - (y_i - s alpha_i)| = %.2e; var in [%.3e, %.3e]; timers %s" % (
    t3 - t2, (t1 - t0) + (t3 - t2), fl / ((t1 - t0) + (t3 - t2)) / 1e12, res, var.min(), var.max(),
    {k: round(v, 1) for k, v in ctx.timers().items() if v}), flush=True)
