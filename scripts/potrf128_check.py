"""potrf128 on a random SPD 128 x 128 block (gpmi_dev_potrf_block) against NumPy: L, and the inverses of the diagonal
16 x 16 tiles that the kernel leaves, transposed, above their diagonals."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussian_process_amd.dist import HipBlockOps
ops = HipBlockOps(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
for n in (128, 256, 512):
    B = rng.standard_normal((n, 2 * n))
    A = B @ B.T / (2 * n) + 0.05 * np.eye(n)
    Ad = torch.from_numpy(A.copy()).to(dev)
    info = torch.full((1,), 2 ** 62, dtype=torch.int64, device=dev)
    ops.potrf_block(Ad, 0, info)
    torch.cuda.synchronize()
    G = Ad.cpu().numpy()
    L = np.linalg.cholesky(A)
    eL = np.abs(np.tril(G) - L).max() / np.abs(L).max()
    eW = 0.0
    for j in range(0, n, 16):
        T = G[j:j + 16, j:j + 16]
        W = np.linalg.inv(np.tril(T))
        Wt_stored = np.triu(T, 1)                       # G[c][r] = W[r][c], r > c
        eW = max(eW, np.abs(Wt_stored - np.triu(W.T, 1)).max() / np.abs(W).max())
    print("n=%d: max|L - chol| / max|L| = %.2e   max|W^T stored - inv| / max|W| = %.2e   info %d"
          % (n, eL, eW, int(info.item()) if int(info.item()) < 2 ** 61 else -1), flush=True)
# a block that is not positive definite: first failing pivot as LAPACK reports it
A = rng.standard_normal((128, 128)); A = A @ A.T / 128 + 0.05 * np.eye(128); A[70, 70] = -1.0
Ad = torch.from_numpy(A.copy()).to(dev); info = torch.full((1,), 2 ** 62, dtype=torch.int64, device=dev)
ops.potrf_block(Ad, 0, info); torch.cuda.synchronize()
try:
    np.linalg.cholesky(A); lap = -1
except np.linalg.LinAlgError:
    import scipy.linalg as sl
    lap = sl.lapack.dpotrf(A, lower=1)[1] - 1
print("not positive definite: info %d, LAPACK %d" % (int(info.item()), lap))
