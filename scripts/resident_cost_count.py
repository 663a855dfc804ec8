"""How the cost grows with the NUMBER of CUs held exclusively: the update GEMM (32768 x 32768 x 2048 lower, random operands)
per form beside k one-wave sleepers of 68 KiB of LDS each (no update workgroup fits beside one), each on a stream of its own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
for form, pers, tk in (("per-tile", 0, 0), ("persistent + stealing", 1, 0), ("ticket", 0, 2)):
    ctx.set_option("gemm_persist", pers); ctx.set_option("gemm_ticket", tk)
    base = ctx.probe_gemm(M, N, K, 1, 32, 4)
    print("%s alone: %.2f TF/s (%.3f ms per launch)" % (form, base[0], base[1]), flush=True)
    for k in (1, 2, 3, 4, 6, 8):
        check(lib.gpmi_probe_resident(ctx._h, 1, 68 * 1024, 64, 1200.0, 0, 0))        # clears the earlier ones (waits), then the first
        for _ in range(k - 1):
            check(lib.gpmi_probe_resident(ctx._h, 1, 68 * 1024, 64, 1200.0, -1, 0))   # more, kept
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, 32, 4)
        print("   beside %2d CU(s) held exclusively: %.2f TF/s (%+.1f %%; the CUs' share is %.1f %%)" % (k, r[0], 100.0 * (r[0] / base[0] - 1.0), -100.0 * k / 256), flush=True)
        time.sleep(0.3)
