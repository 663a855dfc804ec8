"""The trailing update as the multi-rank driver issues it (row map over a rectangle) against the single-GPU
form (lower mode, triangular tile enumeration) on the same triangular region."""
import os, sys
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.dist import HipBlockOps
from gaussian_process_amd._lib import check
ops = HipBlockOps(0)
dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
Cm = torch.zeros(M, M + 544, dtype=torch.float64, device=dev)
A = torch.randn(M, K + 32, dtype=torch.float64, device=dev)
rowmap = torch.tensor([(b + 1) * 128 for b in range(M // 128)], dtype=torch.int32, device=dev)
flops = 2.0 * K * (M * (M + 128) / 2)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def lower():
    check(ops.lib.gpmi_dev_gemm_nt(ops._stream(), ops._p(Cm), Cm.stride(0), ops._p(A), A.stride(0), ops._p(A), A.stride(0),
                                   M, M, K, 1, 0))


t1 = timeit(lower)
t2 = timeit(lambda: ops.gemm_nt_rowmap(Cm[:, :M], A[:, :K], A[:, :K], rowmap, 128))
rowmap_h = rowmap.cpu().numpy()
t3 = timeit(lambda: ops.gemm_nt_rowmap(Cm[:, :M], A[:, :K], A[:, :K], rowmap, 128, rowmap_h))
print("M=%d K=%d: lower mode %.2f ms (%.1f TF/s)   row map, whole rectangle launched %.2f ms (%.1f TF/s)   "
      "row map, live supertiles only %.2f ms (%.1f TF/s)" % (M, K, t1, flops / t1 / 1e9, t2, flops / t2 / 1e9, t3, flops / t3 / 1e9))
