import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for st in (1, 0):
    ctx.set_option("gemm_stagger", st)
    for (M, N, K, lower) in ((16384, 16384, 512, 0), (16384, 16384, 2048, 0), (8192, 8192, 512, 0)):
        for v in (0, 8, 10, 9, 11):
            tf, ms = ctx.probe_gemm(M, N, K, lower, v, 3)
            print("stagger=%d gemm M=%d N=%d K=%d variant=%2d: %.1f TF/s  %.3f ms" % (st, M, N, K, v, tf, ms), flush=True)
for pad in (0, 16, 544, 4096 + 32):
    ctx.set_option("ld_pad", pad)
    for v in (0, 8):
        tf, ms = ctx.probe_gemm(16384, 16384, 512, 0, v, 3)
        print("ld_pad=%d variant=%d: %.1f TF/s" % (pad, v, tf), flush=True)
