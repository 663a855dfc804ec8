"""What a workgroup that merely HOLDS a CU costs the trailing-update GEMM: gpmi_probe_gemm (one 32768 x 32768 x 2048 lower
update, the headline's launch shape) alone, and while one sleeping workgroup is resident -- at normal and at the highest
stream priority, with little LDS (an update workgroup can share its CU) and with 68 KiB (it cannot)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
for K in (2048,):
    base = ctx.probe_gemm(M, N, K, 1, 0, 5)
    print("K=%d alone: %.2f TF/s (%.3f ms per launch)" % (K, base[0], base[1]), flush=True)
    for hp, lds, thr, poll, fences in ((0, 1024, 64, 0, 0), (1, 68 * 1024, 512, 0, 0), (1, 68 * 1024, 512, 64, 0), (1, 68 * 1024, 512, 16, 0),
                                       (1, 68 * 1024, 512, 2, 0), (1, 68 * 1024, 512, 1, 0), (1, 68 * 1024, 512, 2, 1), (0, 68 * 1024, 512, 2, 1),
                                       (1, 1024, 64, 2, 0), (1, 1024, 64, 2, 1),
                                       (1, 21 * 1024, 512, 2, 2), (0, 21 * 1024, 512, 2, 2), (1, 21 * 1024, 128, 2, 2), (1, 21 * 1024, 512, 64, 2),
                                       (1, 68 * 1024, 512, 2, 2), (1, 21 * 1024, 512, 0, 0)):
        check(lib.gpmi_probe_resident(ctx._h, hp, lds, thr, 400.0, poll, fences))
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, 0, 5)
        how = "asleep" if not poll else "polling a flag, s_sleep(%d)%s%s" % (poll, " + fences" if fences & 1 else "", ", other waves parked at a barrier" if fences & 2 else "")
        print("K=%d beside one resident workgroup (priority %s, %2d KiB LDS, %3d threads, %s): %.2f TF/s (%+.1f %%)"
              % (K, "high" if hp else "normal", lds // 1024, thr, how, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
        time.sleep(0.5)
