import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for (M, N, K) in ((16384, 16384, 512), (16384, 16384, 1024)):
    tf, ms = ctx.probe_gemm(M, N, K, 0, 256 + 16, 1)
    print("K=%d stamped build: %.1f TF/s" % (K, tf), flush=True)
