import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np, gp_oracle as O
from gaussian_process_amd import GPContext
N = 131072
X, y, _ = O.synthetic_problem(N, 16, 4)
with GPContext(0) as ctx:
    t0 = time.perf_counter(); ctx.fit(X, y, 1.0, 2.8, 5e-4); print("fit %.2f s" % (time.perf_counter() - t0), flush=True)
    ref = None
    for mode in (1, 2, 2, 1):
        ctx.set_option("trsv_vinv", mode)
        ts = []
        for _ in range(3):
            a = ctx.alpha(); ts.append(ctx.timers()["alpha"])
        if ref is None: ref = a
        print("N=%d trsv_vinv=%d: alpha %.3f ms = %.0f GB/s; vs first %.1e" % (N, mode, min(ts), 4.0 * N * (N + 1) / min(ts) / 1e6, np.abs(a - ref).max() / np.abs(ref).max()), flush=True)
