"""Latency of the panel primitives the distributed driver chains per step (gpmi_dev_potrf_block,
gpmi_dev_trsm_block, small gemm_nt): the critical path of a block step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.dist import HipBlockOps
ops = HipBlockOps(0)
if len(sys.argv) > 1:          # panel_rate.py <trsm_wave option: 0 lane per row, 1 wave per row up to 16384 rows>
    from gaussian_process_amd import GPContext
    GPContext(0).set_option("trsm_wave", int(sys.argv[1]))
dev = torch.device("cuda", 0)
info = torch.full((1,), (1 << 63) - 1, dtype=torch.int64, device=dev)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2] * 1e3


rng = np.random.default_rng(0)
for nb in (128, 512):
    B = rng.standard_normal((nb, nb)); S = B @ B.T + nb * np.eye(nb)
    A0 = torch.from_numpy(S).to(dev); A = torch.empty(nb, nb + 32, dtype=torch.float64, device=dev)
    def f():
        A[:, :nb].copy_(A0); ops.potrf_block(A[:, :nb], 0, info)
    def g():
        A[:, :nb].copy_(A0)
    t = timeit(f) - timeit(g)
    print("potrf_block nb=%4d: %7.1f us  (%.2f TFLOP/s)" % (nb, t, nb ** 3 / 3 / t / 1e6), flush=True)
    L = torch.from_numpy(np.linalg.cholesky(S)).to(dev)
    for m in (512, 4096, 8192, 16384, 65536):
        X = torch.randn(m, nb + 32, dtype=torch.float64, device=dev)
        t = timeit(lambda: ops.trsm_block(L, X[:, :nb]))
        print("   trsm_block m=%6d nb=%4d: %7.1f us  (%.2f TFLOP/s)" % (m, nb, t, m * nb * nb / t / 1e6), flush=True)
for (m, n, k) in ((8192, 512, 512), (2048, 512, 512), (8192, 128, 128), (1024, 256, 256)):
    Cm = torch.randn(m, n + 32, dtype=torch.float64, device=dev)
    Am = torch.randn(m, k + 32, dtype=torch.float64, device=dev)
    Bm = torch.randn(n, k + 32, dtype=torch.float64, device=dev)
    t = timeit(lambda: ops.gemm_nt(Cm[:, :n], Am[:, :k], Bm[:, :k]))
    print("gemm_nt %dx%dx%d: %7.1f us (%.2f TFLOP/s)" % (m, n, k, t, 2.0 * m * n * k / t / 1e6), flush=True)
