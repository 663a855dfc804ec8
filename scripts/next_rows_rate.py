"""SURVEY.md section 8(f) rows on one GPU, timed: f1 posterior-sample factor (post_chol: v^T v SYRK + Cholesky at size n),
f2 LML gradient (K_y^-1 from the resident factor + fused trace), f4 the other covariance functions' matrix builds.
Usage: python scripts/next_rows_rate.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext

ctx = GPContext(0)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


for N in [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768, 65536]:
    n = 1024 if N <= 16384 else 4096
    X, y, Xs = O.synthetic_problem(N, 8, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    t_fit = best(lambda: ctx.factorize(1.0, 2.0, 5e-4))
    t_pred = best(lambda: ctx.predict_resident(want_sd=True))
    t_post = best(lambda: ctx.post_chol(1e-6))          # needs the v of the predict before it
    t_both = best(lambda: (ctx.factorize(1.0, 2.0, 5e-4), ctx.lml_grad()))   # the gradient of a fresh factor each time
    t_grad = t_both - t_fit
    print("N=%6d n=%5d  fit %.4f s  predict %.4f s | f1 post_chol %.4f s (%.1f TFLOP/s on n^2 N + n^3/3) | "
          "f2 lml_grad %.4f s (%.1f TFLOP/s on 2N^3/3; %.2f x the fit)" % (
              N, n, t_fit, t_pred, t_post, (n * n * N + n ** 3 / 3) / t_post / 1e12,
              t_grad, 2.0 * N ** 3 / 3 / t_grad / 1e12, t_grad / t_fit), flush=True)

# f4: full covariance matrices of the other functions (the public cov() call: build + D2H of the matrix is the caller's;
# timed here is the call with a small output so that the build dominates... the matrix must fit the host: N = 8192)
N = 8192
X, y, _ = O.synthetic_problem(N, 8, 4)
X1 = np.ascontiguousarray(X[:, :1])
for kind, p0, p1 in (("rbf", 1.0, 2.0), ("lin", 0.5, 0.0), ("per", 1.0, 3.0)):
    A = X1 if kind == "per" else X
    try:
        t = best(lambda: ctx.cov(kind, A, A, p0, p1), reps=2)
        print("f4 cov(%s) N=%d: %.4f s per call incl. the %.0f MB download" % (kind, N, t, 8.0 * N * N / 1e6), flush=True)
    except Exception as e:  # the kinds a build does not carry fail loudly; say which
        print("f4 cov(%s): %s: %s" % (kind, type(e).__name__, e), flush=True)
