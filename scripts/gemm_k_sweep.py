"""The trailing-update GEMM alone against its K (block width): the per-tile prologue + C read-modify-write share."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for lower in (0, 1):
    for M in (8192, 16384):
        for K in (128, 256, 512, 1024, 2048, 4096):
            tf, ms = ctx.probe_gemm(M, M, K, lower, 0, 5)
            print("lower=%d M=N=%5d K=%4d: %.3f ms, %.1f TF/s" % (lower, M, K, ms, tf), flush=True)
