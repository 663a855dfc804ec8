import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
bad = 0
for (N, d, n, ell, s2, sf) in ((3001, 5, 77, 1.5, 5e-4, 1.0), (5000, 3, 200, 0.9, 1e-3, 1.3), (12289, 8, 130, 2.0, 5e-4, 1.0),
                               (13000, 2, 1000, 0.7, 5e-4, 0.8), (20011, 8, 257, 2.0, 5e-4, 1.0), (8191, 16, 64, 2.8, 1e-4, 1.0),
                               (12416, 1, 300, 0.5, 5e-4, 1.0)):
    X, y, Xs = O.synthetic_problem(N, d, n, seed=N)
    t0 = time.time()
    ref = O.fit_predict_feasible(X, Xs, y, sf, ell, s2)
    t1 = time.time()
    lml = ctx.fit(X, y, sf, ell, s2)
    mu, var = ctx.predict(Xs, want_sd=False)
    al = ctx.alpha()
    e_mu = np.max(np.abs(mu - ref["mu"])); e_var = np.max(np.abs(var - ref["var"]))
    e_lml = abs(lml - ref["lml"]) / abs(ref["lml"]); e_al = np.max(np.abs(al - ref["alpha"])) / np.max(np.abs(ref["alpha"]))
    ok = e_mu <= 1e-8 and e_var <= 1e-9 and e_lml <= 1e-10 and e_al <= 1e-7
    bad += not ok
    print("N=%6d d=%2d n=%4d: dmu %.1e dvar %.1e dlml %.1e dalpha %.1e  (oracle %.1f s) %s" % (N, d, n, e_mu, e_var, e_lml, e_al, t1 - t0, "ok" if ok else "FAIL"), flush=True)
print("failures:", bad)
