"""a5 alone: the backward solve L^T alpha = m at several sizes (ms, GB/s over 8 N (N+1) / 2 bytes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N in [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]:
    X, y, _ = O.synthetic_problem(N, 8, 4)
    ctx.fit(X, y, 1.0, 2.0, 5e-4)
    a = ctx.alpha()
    ts = []
    for _ in range(5):
        a = ctx.alpha(); ts.append(ctx.timers()["alpha"])
    ms = min(ts)
    idx = np.random.default_rng(0).choice(N, 8, replace=False)
    res = max(abs((np.exp(-.125 * ((X - X[i]) ** 2).sum(1)) @ a) + 5e-4 * a[i] - y[i]) for i in idx)
    print("N=%6d: alpha %.3f ms = %.0f GB/s (%.1f %% of 8 TB/s); max residual %.1e" % (N, ms, 4.0 * N * (N + 1) / ms / 1e6,
          4.0 * N * (N + 1) / ms / 1e6 / 80, res), flush=True)
