"""The same question as resident_cost.py for the PER-TILE form of the update GEMM (the one the factorisation uses under
lookahead; resident_cost.py times the persistent form, which draws its tiles from counters): one 32768 x 32768 x 2048 lower
update alone and beside ONE resident workgroup that merely holds a CU."""
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
    base = ctx.probe_gemm(M, N, K, 1, 0, 5)
    print("%s alone: %.2f TF/s (%.3f ms per launch)" % (form, base[0], base[1]), flush=True)
    for hp, lds, thr, poll, fences in ((1, 21 * 1024, 512, 2, 2), (1, 21 * 1024, 512, 0, 0), (1, 1024, 64, 0, 0), (0, 21 * 1024, 512, 0, 0), (1, 100 * 1024, 64, 0, 0)):
        check(lib.gpmi_probe_resident(ctx._h, hp, lds, thr, 400.0, poll, fences))
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, 0, 5)
        how = "asleep" if not poll else "polling, others parked at a barrier"
        print("   %s beside one resident workgroup (priority %s, %3d KiB LDS, %3d threads, %s): %.2f TF/s (%+.1f %%)"
              % (form, "high" if hp else "normal", lds // 1024, thr, how, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
        time.sleep(0.5)
