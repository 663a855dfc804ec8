"""gpmi_lml_batch per triple against the lookahead threshold (option la_min) and the lane count, N below 12288."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
tr = np.array([[l, sf, 5e-4] for l in (1.5, 2.0, 2.5, 3.0) for sf in (0.8, 1.0, 1.2)] * 2)
for N in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 6144, 8192, 10240, 12288, 16384]:
    X, y, _ = O.synthetic_problem(N, 8, 4)
    ctx.set_train(X, y)
    row = []
    ref = None
    for la_min in (12288,):
        for lanes in (0, 2, 3, 4, 5, 6):
            ctx.set_option("la_min", la_min); ctx.set_option("lanes", lanes)
            ctx.lml_batch(tr[:3])
            ts = []
            for rep in range(3):
                t0 = time.perf_counter(); lm, st = ctx.lml_batch(tr); ts.append(time.perf_counter() - t0)
            ref = lm if ref is None else ref
            row.append("la_min=%5d lanes=%d: %.3f ms/triple%s" % (la_min, lanes, min(ts) / len(tr) * 1e3, "" if np.array_equal(lm, ref) else "  DIFFERENT BITS"))
    ctx.set_option("la_min", 12288); ctx.set_option("lanes", 0)
    print("N=%6d\n    " % N + "\n    ".join(row), flush=True)
