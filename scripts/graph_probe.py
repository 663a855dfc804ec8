"""Does a hipGraph replay beat direct launches for a latency-bound launch chain on this stack?
potrf_block(512) (~40 small kernels) captured with torch.cuda.CUDAGraph (the primitives launch on
torch's current stream) against the same calls issued directly."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.dist import HipBlockOps
ops = HipBlockOps(0)
dev = torch.device("cuda", 0)
nb = 512
rng = np.random.default_rng(0)
Bm = rng.standard_normal((nb, nb)); S = Bm @ Bm.T + nb * np.eye(nb)
A0 = torch.from_numpy(S).to(dev)
A = torch.empty(nb, nb + 32, dtype=torch.float64, device=dev)
info = torch.full((1,), (1 << 63) - 1, dtype=torch.int64, device=dev)


def work():
    A[:, :nb].copy_(A0)
    ops.potrf_block(A[:, :nb], 0, info)


work(); torch.cuda.synchronize()
reps = 50
t0 = time.perf_counter()
for _ in range(reps): work()
torch.cuda.synchronize()
direct = (time.perf_counter() - t0) / reps * 1e6
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    work()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    work()
g.replay(); torch.cuda.synchronize()
ref = np.linalg.cholesky(S)
err = np.abs(np.tril(A[:, :nb].cpu().numpy()) - ref).max()
t0 = time.perf_counter()
for _ in range(reps): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / reps * 1e6
print("potrf_block(512): direct launches %.1f us, graph replay %.1f us per factorisation (replay error vs NumPy %.1e)" % (direct, graph, err))
