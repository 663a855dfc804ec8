"""What a resident workgroup's REGISTER footprint costs the per-tile update GEMM: one 32768 x 32768 x 2048 lower update
(six launches) beside one sleeping workgroup of few registers and of ~130 registers per lane, by thread count and LDS."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
for form, pers in (("per-tile", 0), ("persistent + stealing", 1)):
    ctx.set_option("gemm_persist", pers)
    base = ctx.probe_gemm(M, N, K, 1, 32, 6)
    print("%s alone: %.2f TF/s (%.3f ms per launch)" % (form, base[0], base[1]), flush=True)
    for lds, thr, fat in ((21 * 1024, 512, 0), (21 * 1024, 512, 1), (1024, 512, 1), (100 * 1024, 512, 1), (21 * 1024, 256, 1), (21 * 1024, 64, 1), (21 * 1024, 1024, 0)):
        check(lib.gpmi_probe_resident(ctx._h, 1, lds, thr, 500.0, 0, 4 if fat else 0))
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, 32, 6)
        print("   beside one sleeping workgroup (%3d KiB LDS, %4d threads, %s): %.2f TF/s (%+.1f %%)"
              % (lds // 1024, thr, "~130 registers per lane" if fat else "few registers", r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
        time.sleep(0.7)
