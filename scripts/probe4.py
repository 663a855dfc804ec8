import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
ctx.set_option("gemm_dma", 1)
for waves in (4, 8):
    ctx.set_option("gemm_dma_waves", waves)
    for (M, N, K, lower) in ((16384, 16384, 512, 0), (16384, 16384, 1024, 0), (32768, 32768, 1024, 1)):
        tf, ms = ctx.probe_gemm(M, N, K, lower, 0, 5)
        print("waves=%d dma gemm M=%d N=%d K=%d lower=%d: %.1f TF/s  %.3f ms" % (waves, M, N, K, lower, tf, ms), flush=True)
