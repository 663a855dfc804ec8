import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N, d, n in ((512, 1, 100), (2048, 8, 512), (4096, 8, 1024), (8192, 8, 1024), (16384, 8, 4096)):
    X, y, Xs = O.synthetic_problem(N, d, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        lml = ctx.factorize(1.0, 2.0, 5e-4)
        t1 = time.perf_counter()
        mu, var = ctx.predict_resident(False)
        t2 = time.perf_counter()
        ts.append((t2 - t0, t1 - t0, t2 - t1))
    b = min(ts)
    t1 = []
    for rep in range(5):
        t0 = time.perf_counter(); ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False); t1.append(time.perf_counter() - t0)
    print("N=%6d d=%d n=%5d: fit+predict %.3f ms (fit %.3f, predict %.3f); one pass %.3f ms" % (N, d, n, b[0] * 1e3, b[1] * 1e3, b[2] * 1e3, min(t1) * 1e3), flush=True)
