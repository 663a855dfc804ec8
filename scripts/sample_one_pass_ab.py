"""prediction() with its posterior-sample factor: one pass + post_chol (SYRK + second Cholesky) against the augmented
factorisation (gpmi_fit_predict_sample_resident).  python scripts/sample_one_pass_ab.py [N:n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(512, 100), (2048, 512), (4096, 1024), (16384, 1024), (32768, 4096), (65536, 4096)]
for N, n in cases:
    d = 1 if N == 512 else 8
    X, y, Xs = O.synthetic_problem(N, d, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    reps = 6 if N <= 16384 else 3

    def sep():
        lml, mu, sd = ctx.fit_predict_resident(1.0, 2.0, 5e-4)
        return lml, mu, sd, ctx.post_chol(1e-6)

    def aug():
        return ctx.fit_predict_sample_resident(1.0, 2.0, 5e-4, 1e-6)

    res = {}
    for name, f in (("sep", sep), ("aug", aug), ("sep", sep), ("aug", aug)):
        f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = f(); ts.append(time.perf_counter() - t0)
        res.setdefault(name, []).append(min(ts)); res[name + "_out"] = out
    a, b = res["sep_out"], res["aug_out"]
    print("N=%6d n=%5d  one pass + post_chol %s ms   augmented %s ms   |dlml| %.2e rel  max|dmu| %.2e  max|dsd| %.2e  max|dL_| %.2e  max|d(L_ L_^T)| %.2e" % (
        N, n, ["%.3f" % (t * 1e3) for t in res["sep"]], ["%.3f" % (t * 1e3) for t in res["aug"]], abs(a[0] - b[0]) / abs(a[0]),
        np.max(np.abs(a[1] - b[1])), np.nanmax(np.abs(a[2] - b[2])), np.max(np.abs(a[3] - b[3])), np.max(np.abs(a[3] @ a[3].T - b[3] @ b[3].T))), flush=True)
