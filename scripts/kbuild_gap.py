"""Why the K build is slower inside a fit + predict step than back to back: the same launch after (a) nothing,
(b) a host-side idle gap, (c) a trailing-update-sized MFMA GEMM."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import gp_oracle as O
from gaussian_process_amd.dist import HipBlockOps
N, d = 65536, 8
X, y, Xs = O.synthetic_problem(N, d, 16)
ops = HipBlockOps(0)
Xd = torch.from_numpy(X).cuda()
A = torch.empty(N, N + 544, dtype=torch.float64, device="cuda")
Cm = torch.zeros(16384, 16384 + 32, dtype=torch.float64, device="cuda")
Pm = torch.randn(16384, 1024 + 32, dtype=torch.float64, device="cuda")


def kb():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.rbf_rows(Xd, N, d, 0, N, N, 1.0, 2.0, 5e-4, A); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    kb()
print("back to back      :", ["%.3f" % kb() for _ in range(6)], flush=True)
for gap in (0.0005, 0.005, 0.05, 0.5):
    r = []
    for _ in range(5):
        time.sleep(gap); r.append(kb())
    print("after %6.1f ms idle:" % (gap * 1e3), ["%.3f" % v for v in r], flush=True)
r = []
for _ in range(5):
    for _ in range(8):
        ops.gemm_nt(Cm[:, :16384], Pm[:, :1024], Pm[:, :1024])
    r.append(kb())
print("after 8 MFMA GEMMs :", ["%.3f" % v for v in r], flush=True)
r = []
for _ in range(5):
    for _ in range(8):
        ops.gemm_nt(Cm[:, :16384], Pm[:, :1024], Pm[:, :1024])
    torch.cuda.synchronize(); time.sleep(0.001)
    r.append(kb())
print("GEMMs, sync, 1 ms  :", ["%.3f" % v for v in r], flush=True)
