import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, gp_oracle as O
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for N, n in ((16384, 1024), (16384, 4096), (32768, 1024)):
    X, y, Xs = O.synthetic_problem(N, 8, n)
    ctx.set_train(X, y); ctx.set_test(Xs)
    ctx.factorize(1.0, 2.0, 5e-4)
    for la in (1, 0):
        for nb in (0, 512, 256):
            ctx.set_option("lookahead", la); ctx.set_option("nb", nb)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); ctx.predict_resident(False); ts.append(time.perf_counter() - t0)
            print("N=%d n=%d la=%d nb=%d: predict %.2f ms (solve_v %.2f)" % (N, n, la, nb, min(ts) * 1e3, ctx.timers()["solve_v"]), flush=True)
    ctx.set_option("lookahead", 1); ctx.set_option("nb", 0)
