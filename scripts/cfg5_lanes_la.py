"""config 5's size (N=32768): gpmi_lml_batch per triple against the lane count, with each lane's lookahead on and off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y, _ = O.synthetic_problem(N, 8, 4)
ctx.set_train(X, y)
tr = np.array([[l, sf, 5e-4] for l in (1.5, 2.0, 2.5, 3.0) for sf in (0.8, 1.0, 1.2)])
ref = None
for la in (1, 0):
    for lanes in (2, 3, 4, 5, 6):
        ctx.set_option("lookahead", la); ctx.set_option("lanes", lanes)
        ctx.lml_batch(tr[:lanes])
        t0 = time.perf_counter(); lm, st = ctx.lml_batch(tr); dt = time.perf_counter() - t0
        ref = lm if ref is None else ref
        print("N=%d lookahead=%d lanes=%d: %.4f s per triple (%.1f TFLOP/s)%s" % (N, la, lanes, dt / len(tr), N ** 3 / 3 * len(tr) / dt / 1e12,
              "" if np.array_equal(lm, ref) else "  DIFFERENT BITS"), flush=True)
