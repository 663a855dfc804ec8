"""One warm fit + predict at a given size, for rocprofv3 --kernel-trace (scripts/trace_size.sh): the second call is the
one to read.   python scripts/trace_one.py N n [option=value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N, n = int(sys.argv[1]), int(sys.argv[2])
ctx = GPContext(0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
X, y, Xs = O.synthetic_problem(N, 8, n)
ctx.set_train(X, y); ctx.set_test(Xs)
for rep in range(2):
    if os.environ.get("TRACE_ONE_PASS") == "1":       # prediction() in one pass (the bench's step)
        t0 = time.perf_counter(); lml, mu, var = ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False); t1 = time.perf_counter()
        print("rep %d: one pass %.3f ms  lml %.6f" % (rep, (t1 - t0) * 1e3, lml), flush=True)
        continue
    t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
    mu, var = ctx.predict_resident(False); t2 = time.perf_counter()
    print("rep %d: fit %.3f ms predict %.3f ms  lml %.6f" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, lml), flush=True)
