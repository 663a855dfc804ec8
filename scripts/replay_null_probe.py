"""What do the replay's stand-in copies cost the replayed rank?  Rank r of 8 at N=65536 replayed normally and with NULL
delivery (GPMI_REPLAY_NULL=1: the absent ranks' panel columns / diagonal blocks / v blocks are not copied in; results are
garbage, a failed pivot is ignored, the timing is the rank's own kernels and Python alone).  python scripts/replay_null_probe.py [rank]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.replay import ReplaySource, replay_rank
r = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, d, n, G = 65536, 8, 4096, 8
rng = np.random.default_rng(20240531)
X = rng.uniform(-1, 1, (N, d)); y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N); Xs = rng.uniform(-1, 1, (n, d))
torch.cuda.set_device(0)
src = ReplaySource(0, 1024, X, y, Xs, 1.0, 2.0, 5e-4)
for null in ("0", "1", "0", "1"):
    os.environ["GPMI_REPLAY_NULL"] = null
    gp = replay_rank(0, src, r, G, X, y, Xs)
    ts = []
    for k in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            gp.factorize(1.0, 2.0, 5e-4)
        except np.linalg.LinAlgError:
            gp.have_factor = True
        t1 = time.perf_counter()
        gp.predict_resident(want_sd=False)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    b = min(ts)
    print("rank %d null=%s: fit + predict %.1f ms (fit %.1f predict %.1f)" % (r, null, b[0], b[1], b[2]), flush=True)
    del gp
