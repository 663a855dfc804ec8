"""fit + alpha + predict with lookahead from fewer columns than the 12288 it was set at in round 2 (option la_min), as one
pass and as two calls.   python scripts/la_min_sweep.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [1024, 1536, 2048, 3072, 4096, 6144, 8192, 10240, 12288]:
    X, y, Xs = O.synthetic_problem(N, 8, 1024)
    ctx.set_train(X, y); ctx.set_test(Xs)
    row, outs = [], []
    for la_min in (1 << 30, 12288, 1024):
        ctx.set_option("la_min", la_min)
        t1, t2 = [], []
        for rep in range(7):
            t0 = time.perf_counter()
            lml, mu, var = ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False)
            a = ctx.alpha()
            t1.append(time.perf_counter() - t0)
        for rep in range(7):
            t0 = time.perf_counter()
            lml2 = ctx.factorize(1.0, 2.0, 5e-4); a2 = ctx.alpha(); mu2, var2 = ctx.predict_resident(want_sd=False)
            t2.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ctx.factorize(1.0, 2.0, 5e-4); tf = time.perf_counter() - t0
        outs.append((lml, mu, var, a, lml2, mu2, var2))
        row.append("la_min=%-10d one pass %.3f  two calls %.3f  fit alone %.3f ms" % (la_min, min(t1[1:]) * 1e3, min(t2[1:]) * 1e3, tf * 1e3))
    ctx.set_option("la_min", 12288)
    same = all(o[0] == outs[0][0] and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2]) and np.array_equal(o[3], outs[0][3])
               and o[4] == outs[0][0] and np.array_equal(o[5], outs[0][1]) for o in outs)
    print("N=%6d  same bits everywhere: %s\n    " % (N, same) + "\n    ".join(row), flush=True)
