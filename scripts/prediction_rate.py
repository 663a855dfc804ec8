"""The drop-in prediction(X_train, X_test, y_train, 'rbf', l, num_fun) end to end (uploads, one augmented Cholesky, samples
formed on the device, downloads) against the same through separate calls with L_ brought to the host.
   python scripts/prediction_rate.py [N:n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
from gaussian_process_amd import GP_regression as G
ctx = GPContext(0)
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(512, 100), (2048, 512), (4096, 1024), (8192, 2048), (16384, 1024), (16384, 4096), (65536, 4096)]
for N, n in cases:
    d = 1 if N == 512 else 8
    X, y, Xs = O.synthetic_problem(N, d, n)
    ell = 1.0 if d == 1 else 2.0
    reps = 5 if N <= 16384 else 2

    def dropin():
        np.random.seed(5)
        return G.prediction(X, Xs, y, 'rbf', ell, 10, ctx=ctx)

    def separate():
        np.random.seed(5)
        ctx.fit(X, y, 1.0, ell, 5e-4)
        mu, sd = ctx.predict(Xs)
        L_ = ctx.post_chol(1e-6)
        return mu, sd, mu.reshape(-1, 1) + L_ @ np.random.normal(size=(n, 10))

    res = {}
    for name, f in (("separate", separate), ("dropin", dropin), ("separate", separate), ("dropin", dropin)):
        f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
        res.setdefault(name, []).append(min(ts)); res[name + "_out"] = out
    a, b = res["separate_out"], res["dropin_out"]
    print("N=%6d n=%5d  fit + predict + post_chol + host dot %s ms   prediction() %s ms   max|dmu| %.2e  max|dsd| %.2e  max|df_post| %.2e" % (
        N, n, ["%.3f" % (t * 1e3) for t in res["separate"]], ["%.3f" % (t * 1e3) for t in res["dropin"]],
        np.max(np.abs(a[0] - b[0])), np.nanmax(np.abs(a[1] - b[1])), np.max(np.abs(a[2] - b[2]))), flush=True)
