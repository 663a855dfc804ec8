"""config 5's batch (N=32768 triples through gpmi_lml_batch) against the lane count and the per-lane lookahead:
   python scripts/cfg5_opts.py [N] [triples]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
X, y, _ = O.synthetic_problem(N, 8, 4)
triples = np.array([[l, sf, s2] for l in (1., 2., 3., 4.) for sf in (.5, 1., 1.5, 2.) for s2 in (1e-4, 5e-4, 1e-3, 5e-3)])[:T]
ref = None
for lanes, la in ((2, 1), (1, 1), (3, 1), (4, 1), (3, 0), (4, 0), (2, 1)):
    with GPContext(0) as ctx:
        ctx.set_option("lanes", lanes); ctx.set_option("lookahead", la)
        ctx.set_train(X, y)
        ctx.lml_batch(triples[:lanes])
        t0 = time.perf_counter(); lml, st = ctx.lml_batch(triples); dt = time.perf_counter() - t0
        if ref is None:
            ref = lml
        print("N=%d lanes=%d lookahead=%d: %d triples in %.2f s (%.4f s each, %.1f TFLOP/s on N^3/3 each); same bits as the default: %s"
              % (N, lanes, la, T, dt, dt / T, T * N ** 3 / 3 / dt / 1e12, np.array_equal(lml, ref)), flush=True)
