"""cfg5-style batch (gpmi_lml_batch): wall per triple against the number of factorisations in flight."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
T = 12
ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [512, 2048, 8192, 16384, 32768]:
    X, y, _ = O.synthetic_problem(N, 8, 4)
    triples = np.array([[1.0 + 0.25 * t, 1.0, 5e-4] for t in range(T)])
    ctx.set_train(X, y)
    ref = None
    line = "N=%6d:" % N
    for lanes in (1, 2, 3, 4, 6):
        ctx.set_option("lanes", lanes)
        ctx.lml_batch(triples)                      # creates the lanes / warms up
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); l, _ = ctx.lml_batch(triples); best = min(best, time.perf_counter() - t0)
        if ref is None: ref = l
        assert np.array_equal(l, ref)
        line += "  lanes=%d %.3f ms" % (lanes, best / T * 1e3)
    print(line, flush=True)
