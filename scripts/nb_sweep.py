"""fit + predict wall against the outer block width (environment GPMI_NB is read at context creation):
   python scripts/nb_sweep.py N n nb1 nb2 ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N, n = int(sys.argv[1]), int(sys.argv[2])
X, y, Xs = O.synthetic_problem(N, 8, n)
for nb in [int(a) for a in sys.argv[3:]]:
    if nb:
        os.environ["GPMI_NB"] = str(nb)
    else:
        os.environ.pop("GPMI_NB", None)
    with GPContext(0) as ctx:
        ctx.set_train(X, y); ctx.set_test(Xs)
        best = None
        for rep in range(4):
            t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
            tm = ctx.timers()
            mu, var = ctx.predict_resident(False); t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1, tm)
        print("N=%d n=%d nb=%4d: total %.2f ms (fit %.2f predict %.2f) chol %.2f panel-stream %.2f trail %.2f lml %.9f"
              % (N, n, nb, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3]["chol"], best[3]["chol_panel"], best[3]["chol_trail"], lml), flush=True)
