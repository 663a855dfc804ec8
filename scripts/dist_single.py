"""DistGP with world_size 1 (no collectives) vs the single-GPU C++ driver: orchestration overhead."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import gp_oracle as O
from gaussian_process_amd import GPContext
from gaussian_process_amd.dist import DistGP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = 4096
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
X, y, Xs = O.synthetic_problem(N, 8, n)
for nb, la in [(int(a.split(',')[0]), int(a.split(',')[1])) for a in (sys.argv[2:] or ['512,2', '512,1', '512,0', '1024,2', '1024,1', '1024,0'])]:
    if True:
        gp = DistGP(0, nb=nb, lookahead=la)
        gp.set_train(X, y); gp.set_test(Xs)
        best = None
        for rep in range(3):
            t0 = time.perf_counter(); lml = gp.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
            mu, var = gp.predict_resident(False); t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]: best = (t2 - t0, t1 - t0, t2 - t1)
        print("DistGP G=1 nb=%d lookahead=%d: total %.4f fit %.4f predict %.4f lml %.6f" % (nb, la, *best, lml), flush=True)
        del gp
ctx = GPContext(0)
ctx.set_train(X, y); ctx.set_test(Xs)
best = None
for rep in range(3):
    t0 = time.perf_counter(); lml = ctx.factorize(1.0, 2.0, 5e-4); t1 = time.perf_counter()
    mu2, var2 = ctx.predict_resident(False); t2 = time.perf_counter()
    if best is None or t2 - t0 < best[0]: best = (t2 - t0, t1 - t0, t2 - t1)
print("GPContext: total %.4f fit %.4f predict %.4f lml %.6f  |dmu| %.1e" % (*best, lml, np.abs(mu - mu2).max()))
dist.destroy_process_group()
