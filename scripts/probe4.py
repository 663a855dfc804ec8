"""LDS-DMA GEMM stand-alone: full kernel and its timing-only ablations (dbg build), 8 and 4 waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
ctx.set_option("gemm_dma", 1)
for waves in (8, 4):
    ctx.set_option("gemm_dma_waves", waves)
    for K in (1024, 2048):
        for v in (0, 8, 9, 10, 11):
            tf, ms = ctx.probe_gemm(16384, 16384, K, 0, 256 + v if v else 0, 5)
            print("waves=%d K=%d variant=%2d: %.1f TF/s" % (waves, K, v, tf), flush=True)
