import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
ctx.set_option("gemm_dma", 1)
for (M, N, K, lower) in ((16384, 16384, 512, 0), (16384, 16384, 2048, 0)):
    for v in (0, 256 + 8, 256 + 1, 256 + 1 + 8, 256 + 1 + 2 + 8, 256 + 2 + 8):
        tf, ms = ctx.probe_gemm(M, N, K, lower, v, 5)
        print("dma gemm M=%d N=%d K=%d variant=%d: %.1f TF/s  %.3f ms" % (M, N, K, v & 255, tf, ms), flush=True)
