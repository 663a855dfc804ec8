"""f2 timing: fit + LML gradient at growing N (d = 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768]:
    X, y, _ = O.synthetic_problem(N, 8, 4)
    ctx.set_train(X, y)
    ctx.factorize(1.0, 2.0, 5e-4); ctx.lml_grad()
    t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
    dl, ds = ctx.lml_grad(); t2 = time.perf_counter()
    fl = 2.0 * N ** 3 / 3
    print("N=%6d fit %.4f s  grad %.4f s (%.1f TFLOP/s on 2N^3/3)  dLML/dl %.6e dLML/dsigma %.6e" % (
        N, t1 - t0, t2 - t1, fl / (t2 - t1) / 1e12, dl, ds), flush=True)
