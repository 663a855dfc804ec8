"""The trailing-update GEMM with resident workgroups that chain the K loops of consecutive tiles (option gemm_persist)
against one workgroup per tile, full and lower mode, K = 256 .. 2048; and a fit + predict with both."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for lower in (0, 1):
    for M in (8192, 16384):
        for K in (256, 512, 1024, 2048):
            r = []
            for pers in (0, 1):
                ctx.set_option("gemm_persist", pers)
                tf, ms = ctx.probe_gemm(M, M, K, lower, 0, 5)
                r.append((ms, tf))
            print("lower=%d M=N=%5d K=%4d: per-tile %.3f ms %.1f TF/s | persistent %.3f ms %.1f TF/s (%+.1f %%)"
                  % (lower, M, K, r[0][0], r[0][1], r[1][0], r[1][1], 100 * (r[0][0] / r[1][0] - 1)), flush=True)
for N, n in ((8192, 512), (16384, 1024), (32768, 4096)) + (((65536, 4096),) if "big" in sys.argv else ()):
    X, y, Xs = O.synthetic_problem(N, 8, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    out = []
    for pers in (0, 1):
        ctx.set_option("gemm_persist", pers)
        best = None
        for _ in range(3):
            import time
            t0 = time.perf_counter()
            lml = ctx.factorize(1.0, 2.0, 5e-4)
            mu, var = ctx.predict_resident(False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append((best, lml, mu.copy(), var.copy()))
    print("N=%6d n=%d: per-tile %.2f ms | persistent %.2f ms (%+.1f %%) | dlml %.1e dmu %.1e dvar %.1e"
          % (N, n, out[0][0] * 1e3, out[1][0] * 1e3, 100 * (out[0][0] / out[1][0] - 1), abs(out[0][1] - out[1][1]),
             np.abs(out[0][2] - out[1][2]).max(), np.abs(out[0][3] - out[1][3]).max()), flush=True)
