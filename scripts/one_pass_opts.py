"""prediction() in one pass (+ alpha) against context options that were tuned on the two-call form: shallow_min, ramp.
   python scripts/one_pass_opts.py N n"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
N, n = int(sys.argv[1]), int(sys.argv[2])
X, y, Xs = O.synthetic_problem(N, 8, n)
ctx = GPContext(0)
ctx.set_train(X, y); ctx.set_test(Xs)
ref = None
for name, vals, default in (("shallow_min", (6144, 0, 2048, 4096, 8192, 12288), 6144), ("ramp", (0, 2, 2 + 16 * 2, 2 + 16 * 4, 6), 0)):
    for v in vals:
        ctx.set_option(name, v)
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            lml, mu, var = ctx.fit_predict_resident(1.0, 2.0, 5e-4, want_sd=False)
            ctx.alpha()
            ts.append(time.perf_counter() - t0)
        ref = (lml, mu) if ref is None else ref
        print("N=%d n=%d %s=%d: %.2f ms (runs %s)  %s" % (N, n, name, v, min(ts[1:]) * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts[1:]),
              "same bits" if (lml == ref[0] and np.array_equal(mu, ref[1])) else "lml %.12g (first %.12g)" % (lml, ref[0])), flush=True)
    ctx.set_option(name, default)
