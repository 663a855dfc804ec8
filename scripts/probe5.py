"""Per-tile phase clocks of the trailing-update GEMM (stamped probe build): prologue / K loop / C loads / C stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for (M, N, K) in ((16384, 16384, 256), (16384, 16384, 512), (16384, 16384, 1024), (16384, 16384, 2048)):
    tf0, ms0 = ctx.probe_gemm(M, N, K, 0, 0, 3)
    tf, ms = ctx.probe_gemm(M, N, K, 0, 256 + 16, 1)
    print("K=%d: %.3f ms %.1f TF/s; stamped build: %.1f TF/s" % (K, ms0, tf0, tf), flush=True)
