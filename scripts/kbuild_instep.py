"""K build inside factorize (the bench's in-step figure) for a few rbf_blocks settings, N=65536 d=8."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gp_oracle as O
from gaussian_process_amd import GPContext
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X, y, Xs = O.synthetic_problem(N, 8, 256)
ctx = GPContext(0)
ctx.set_train(X, y); ctx.set_test(Xs)
T = N // 128
bytes_ = 8.0 * 128 * 128 * T * (T + 1) / 2 + 16.0 * N * 8
for blocks in [int(a) for a in sys.argv[2:]] or [16384, 4096, 2048, 1024]:
    ctx.set_option("rbf_blocks", blocks)
    ks = []
    for rep in range(3):
        ctx.factorize(1.0, 2.0, 5e-4)
        ks.append(ctx.timers()["kbuild"])
        ctx.predict_resident(False)
    print("rbf_blocks %6d: kbuild in step %s ms -> %.3f of 8 TB/s (best)" % (blocks, ["%.3f" % k for k in ks], bytes_ / min(ks) / 1e6 / 8000), flush=True)
