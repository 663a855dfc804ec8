"""Ticket form of the trailing-update kernel against the per-tile launch: standalone GEMM rates (gpmi_probe_gemm) and
fit + predict walls at the mid sizes and the headline, per (gemm_ticket, gemm_reserve).  Same bits expected (lml printed
with 12 digits).   python scripts/ticket_ab.py [sizes...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
sizes = [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536]
combos = [(0, 0), (1, 0), (1, 1), (1, 2), (1, 4)]
with GPContext(0) as ctx:
    for (M, N, K, lower) in ((32768, 32768, 2048, 1), (16384, 16384, 1024, 1), (8192, 8192, 1024, 1), (24576, 1024, 1024, 0)):
        for tk, rs in ((0, 0), (2, 0), (2, 1), (2, 2)):
            ctx.set_option("gemm_persist", 0); ctx.set_option("gemm_ticket", tk); ctx.set_option("gemm_reserve", rs)
            tf, ms = ctx.probe_gemm(M, N, K, lower, 0, 3)
            print("probe %dx%dx%d lower=%d ticket=%d reserve=%d: %.2f TF/s %.3f ms" % (M, N, K, lower, tk, rs, tf, ms), flush=True)
    ctx.set_option("gemm_persist", 1)
    for N in sizes:
        n = 1024 if N <= 16384 else 4096
        X, y, Xs = O.synthetic_problem(N, 8, n)
        ctx.set_train(X, y); ctx.set_test(Xs)
        for tk, rs in combos:
            ctx.set_option("gemm_ticket", tk); ctx.set_option("gemm_reserve", rs)
            best = None
            for rep in range(4 if N <= 32768 else 2):
                t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
                tm = ctx.timers()
                mu, var = ctx.predict_resident(False); t2 = time.perf_counter()
                if best is None or t2 - t0 < best[0]:
                    best = (t2 - t0, t1 - t0, t2 - t1, tm)
            print("N=%d n=%d ticket=%d reserve=%d: total %.2f ms (fit %.2f predict %.2f) chol %.2f panel-stream %.2f trail %.2f lml %.12f mu0 %.15e"
                  % (N, n, tk, rs, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3]["chol"], best[3]["chol_panel"], best[3]["chol_trail"], lml, mu[0]), flush=True)
