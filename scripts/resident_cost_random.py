"""resident_cost_pertile.py with pseudo-random operands (gpmi_probe_gemm variant 32): real data toggles the matrix pipe's
inputs and puts the chip at its power limit, constant operands do not."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
ctx.set_option("gemm_persist", 0)
for variant, what in ((0, "constant operands"), (32, "random operands")):
    base = ctx.probe_gemm(M, N, K, 1, variant, 8)
    print("per-tile, %s, alone: %.2f TF/s (%.3f ms per launch)" % (what, base[0], base[1]), flush=True)
    for hp, lds, thr, poll, fences in ((1, 21 * 1024, 512, 2, 2), (1, 21 * 1024, 512, 0, 0), (1, 1024, 64, 2, 0), (1, 1024, 64, 64, 0)):
        check(lib.gpmi_probe_resident(ctx._h, hp, lds, thr, 600.0, poll, fences))
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, variant, 8)
        how = "asleep" if not poll else "polling s_sleep(%d)%s" % (poll, ", others parked at a barrier" if fences & 2 else "")
        print("   beside one resident workgroup (%3d KiB LDS, %3d threads, %s): %.2f TF/s (%+.1f %%)" % (lds // 1024, thr, how, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
        time.sleep(0.8)
