"""a5 alone: the backward solve L^T alpha = m at several sizes (ms, GB/s over 8 N (N+1) / 2 bytes), with the
inverted 128 x 128 diagonal blocks (option trsv_vinv = 1: one launch per block; 2: ONE launch, column blocks chained through the
solution vector; the first call after a fit pays the inversion launch) and with
the 16 x 16 rounds of the second generation (trsv_vinv = 0)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]:
    X, y, _ = O.synthetic_problem(N, 8, 4)
    ref = None
    for vinv in (0, 1, 2):
        ctx.set_option("trsv_vinv", vinv)
        first = []
        for _ in range(3):
            ctx.fit(X, y, 1.0, 2.0, 5e-4)
            a = ctx.alpha(); first.append(ctx.timers()["alpha"])
        ts = []
        for _ in range(5):
            a = ctx.alpha(); ts.append(ctx.timers()["alpha"])
        ms, ms1 = min(ts), min(first)
        idx = np.random.default_rng(0).choice(N, 8, replace=False)
        res = max(abs((np.exp(-.125 * ((X - X[i]) ** 2).sum(1)) @ a) + 5e-4 * a[i] - y[i]) for i in idx)
        if ref is None:
            ref = a.copy()
        print("N=%6d vinv=%d: alpha %.3f ms after a fit, %.3f ms repeated = %.0f GB/s (%.1f %% of 8 TB/s); max residual %.1e; "
              "vs vinv=0 %.1e of max|alpha|" % (N, vinv, ms1, ms, 4.0 * N * (N + 1) / ms / 1e6, 4.0 * N * (N + 1) / ms / 1e6 / 80,
                                               res, np.abs(a - ref).max() / np.abs(ref).max()), flush=True)
