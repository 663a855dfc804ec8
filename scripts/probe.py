"""Micro-benchmarks: fp64 MFMA issue rate / clock, GEMM kernel ablations, HBM write."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext  # noqa: E402

ctx = GPContext(0)
for bpc in (1, 2, 4):
    for nacc in (4, 8, 16):
        tf, ghz, cyc = ctx.probe_mfma_f64_ex(bpc, nacc, 4096)
        print("mfma probe: %d waves/SIMD nacc %2d -> %.1f TF/s  clock %.2f GHz  %.1f cycles/MFMA/SIMD"
              % (bpc, nacc, tf, ghz, cyc), flush=True)
for (M, N, K, lower) in ((16384, 16384, 512, 0), (16384, 16384, 1024, 0), (32768, 32768, 512, 1)):
    for v in (0, 4, 8, 1 | 8, 3 | 8, 1, 3):
        tf, ms = ctx.probe_gemm(M, N, K, lower, v, 3)
        print("gemm M=%d N=%d K=%d lower=%d variant=%2d: %.1f TF/s  %.3f ms" % (M, N, K, lower, v, tf, ms),
              flush=True)
for gb in (1, 4):
    print("hbm write %d GiB: %.0f GB/s" % (gb, ctx.probe_hbm_write(gb << 30)))
