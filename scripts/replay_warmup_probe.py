"""How many steps does a freshly started rank need to reach its steady state?  Rank r of G replayed alone (replay.py) in a
fresh process, no single-GPU reference run first: wall per step for the first `steps` steps.
  python scripts/replay_warmup_probe.py [rank] [G] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.replay import ReplaySource, replay_rank
r, G, steps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 0), (2, 8), (3, 14)))
pad_mb = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0        # a dummy allocation in front of the rank's buffers (address shift)
src_la = int(sys.argv[5]) if len(sys.argv) > 5 else 2            # lookahead of the SOURCE factorisation (0: it creates no streams)
ndummy = int(sys.argv[6]) if len(sys.argv) > 6 else 0            # high-priority streams created (and kept) before the rank's own
N, d, n = 65536, 8, 4096
rng = np.random.default_rng(20240531)
X = rng.uniform(-1, 1, (N, d)); y = np.sin(0.9 * X.sum(1)) + np.sqrt(5e-4) * rng.standard_normal(N); Xs = rng.uniform(-1, 1, (n, d))
torch.cuda.set_device(0)
t0 = time.perf_counter()
src = ReplaySource(0, 1024, X, y, Xs, 1.0, 2.0, 5e-4, lookahead=src_la)
torch.cuda.synchronize()
print("source factorisation: %.2f s" % (time.perf_counter() - t0), flush=True)
pad = torch.empty(int(pad_mb * (1 << 20)), dtype=torch.uint8, device="cuda") if pad_mb > 0 else None
lo, hi = torch.cuda.Stream.priority_range()
dummies = [torch.cuda.Stream(priority=hi) for _ in range(ndummy)]
gp = replay_rank(0, src, r, G, X, y, Xs)
print("source lookahead %d, %d dummy high-priority streams, GPU_MAX_HW_QUEUES=%s" % (src_la, ndummy, os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
print("pad %.1f MB; addresses: A %#x  Lkk %#x  Lk1 %#x  send %#x  Pbuf %#x %#x  V %#x  src.A %#x" % (
    pad_mb, gp.A.data_ptr(), gp.Lk[0].data_ptr(), gp.Lk[1].data_ptr(), gp.send.data_ptr(), gp.Pbuf[0].data_ptr(), gp.Pbuf[1].data_ptr(),
    gp.V.data_ptr(), src.gp.A.data_ptr()), flush=True)
for k in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gp.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
    gp.alpha(); t2 = time.perf_counter()
    gp.predict_resident(want_sd=False)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if k >= 2 and k < steps - 1:
        continue
    print("step %2d: %.1f ms (fit %.1f alpha %.1f predict %.1f)  mem reserved %.1f GB" % (k, (t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, torch.cuda.memory_reserved() / 1e9), flush=True)
