import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N, ell, s in ((8192, 2.0, 5e-4), (8192, 4.0, 5e-4), (8192, 6.0, 5e-4), (8192, 4.0, 1e-6)):
    X, y, Xs = O.synthetic_problem(N, 8, 256)
    ref = O.fit_predict_feasible(X, Xs, y, 1.0, ell, s)
    lml = ctx.fit(X, y, 1.0, ell, s)
    mu, var = ctx.predict(Xs, want_sd=False)
    print("N=%d ell=%.1f s=%.0e: |dmu| %.2e  |dvar| %.2e  dlml_rel %.2e  max|alpha| %.2e" % (
        N, ell, s, np.abs(mu - ref["mu"]).max(), np.abs(var - ref["var"]).max(), abs(lml - ref["lml"]) / abs(ref["lml"]),
        np.abs(ref["alpha"]).max()), flush=True)
