"""A/B of the panel kernels (option panel_fused) at several sizes: fit + predict wall and stage timers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
sizes = [int(a) for a in sys.argv[1:]] or [2048, 8192, 16384, 32768, 65536]
for N in sizes:
    n = 1024 if N <= 16384 else 4096
    X, y, Xs = O.synthetic_problem(N, 8, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    for fused in (0, 1):
        ctx.set_option("panel_fused", fused)
        best = None
        for rep in range(3):
            t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
            tm = ctx.timers()
            mu, var = ctx.predict_resident(False); t2 = time.perf_counter()
            tm2 = ctx.timers()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1, tm, tm2)
        tot, tf, tp, tm, tm2 = best
        print("N=%6d n=%d fused=%d: total %.2f ms (fit %.2f predict %.2f) | chol %.2f panel %.2f trail %.2f solve_v %.2f | lml %.9f"
              % (N, n, fused, tot * 1e3, tf * 1e3, tp * 1e3, tm["chol"], tm["chol_panel"], tm["chol_trail"], tm2["solve_v"], lml), flush=True)
ctx.set_option("panel_fused", 1)
