"""Register-footprint threshold: the per-tile update GEMM (32768 x 32768 x 2048 lower, random operands) beside one sleeping
resident workgroup of 8 waves with ~106 / ~130 / ~138 / ~146 registers per lane."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
ctx.set_option("gemm_persist", 0)
base = ctx.probe_gemm(32768, 32768, 2048, 1, 32, 6)
print("per-tile alone: %.2f TF/s (%.3f ms per launch)" % base, flush=True)
for thr, fat, what in ((512, 3, "108"), (512, 0, "132"), (512, 1, "140"), (512, 2, "148"), (256, 2, "148 (4 waves)"), (512, 1, "140 again")):
    check(lib.gpmi_probe_resident(ctx._h, 1, 21 * 1024, thr, 500.0, 0, 4 | (fat << 5)))
    time.sleep(0.02)
    r = ctx.probe_gemm(32768, 32768, 2048, 1, 32, 6)
    print("   beside one sleeping workgroup of %d threads x %s registers per lane: %.2f TF/s (%+.1f %%)" % (thr, what, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
    time.sleep(0.7)
