"""fit + predict wall for combinations of context options:  python scripts/opt_combo.py N n "opt=v,opt=v" ["opt=v" ...]
(an empty string "" = defaults).  Options set by one combination are put back to the value they had before it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
DEFAULTS = {"potrf_server": 0, "gemm_balance": 1, "gemm_ticket": 0, "gemm_reserve": 0, "gemm_persist": 1, "lookahead": 1, "shallow_min": 6144, "nb": 0}
N, n = int(sys.argv[1]), int(sys.argv[2])
X, y, Xs = O.synthetic_problem(N, 8, n)
with GPContext(0) as ctx:
    ctx.set_train(X, y); ctx.set_test(Xs)
    for combo in sys.argv[3:]:
        kv = dict((a.split("=")[0], int(a.split("=")[1])) for a in combo.split(",") if a)
        for k, v in kv.items():
            ctx.set_option(k, v)
        best = None
        for rep in range(5 if N <= 32768 else 2):
            t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
            tm = ctx.timers()
            mu, var = ctx.predict_resident(False); t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1, tm)
        print("N=%d n=%d [%s]: total %.2f ms (fit %.2f predict %.2f) chol %.2f panel-stream %.2f trail %.2f lml %.12f"
              % (N, n, combo, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3]["chol"], best[3]["chol_panel"], best[3]["chol_trail"], lml), flush=True)
        for k in kv:
            ctx.set_option(k, DEFAULTS.get(k, 0))
