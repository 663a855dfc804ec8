"""K build alone, back to back (steady clocks): gpmi_dev_rbf_rows over all lower tiles."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import gp_oracle as O
from gaussian_process_amd.dist import HipBlockOps
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = 20
X, y, Xs = O.synthetic_problem(N, d, 16)
ops = HipBlockOps(0)
if len(sys.argv) > 3:
    from gaussian_process_amd import GPContext
    GPContext(0).set_option("rbf_blocks", int(sys.argv[3]))
Xd = torch.from_numpy(X).cuda()
A = torch.empty(N, N + 544, dtype=torch.float64, device="cuda")
T = N // 128
bytes_ = 8.0 * 128 * 128 * T * (T + 1) / 2 + 16.0 * N * d
for sigma in (1.0, 1.3):
    ops.rbf_rows(Xd, N, d, 0, N, N, sigma, 2.0, 5e-4, A)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        ops.rbf_rows(Xd, N, d, 0, N, N, sigma, 2.0, 5e-4, A)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    print("sigma %.1f N=%d d=%d: first %.3f ms  min %.3f ms  median %.3f ms -> %.2f TB/s (median) %.2f TB/s (best)" % (
        sigma, N, d, ms[0], min(ms), sorted(ms)[reps // 2], bytes_ / sorted(ms)[reps // 2] / 1e9, bytes_ / min(ms) / 1e9), flush=True)
