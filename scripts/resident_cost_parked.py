"""Is it the waves PARKED AT A BARRIER?  The per-tile update GEMM (32768 x 32768 x 2048 lower, random operands) beside one
resident workgroup of 8 waves x ~130 registers whose waves 1-7 (a) sleep in a loop, (b) wait at a workgroup barrier for
wave 0, (c) poll an LDS word with s_sleep between reads."""
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
for thr, park, what in ((512, 0, "every wave sleeps in a loop"), (512, 1, "waves 1-7 wait at a workgroup barrier"), (512, 2, "waves 1-7 poll an LDS word"),
                        (128, 1, "2 waves: wave 1 waits at a barrier"), (512, 1, "waves 1-7 wait at a workgroup barrier (again)")):
    check(lib.gpmi_probe_resident(ctx._h, 1, 21 * 1024, thr, 500.0, 0, 4 | (park << 3)))
    time.sleep(0.02)
    r = ctx.probe_gemm(32768, 32768, 2048, 1, 32, 6)
    print("   beside one resident workgroup of %d threads x ~130 registers, %s: %.2f TF/s (%+.1f %%)" % (thr, what, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
    time.sleep(0.7)
