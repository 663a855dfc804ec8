"""What a RESIDENT workgroup costs the update GEMM -- measured with the sleeper actually resident (until round 4 the probe
destroyed the sleeper's stream right after launching it, and hipStreamDestroy waits: everything timed "beside" a sleeper ran
alone).  One 32768 x 32768 x 2048 lower update (random operands, six launches) per GEMM form beside ONE sleeping workgroup:
by threads, registers per lane, LDS, and whether an update workgroup (96 KiB of LDS, 8 waves x 144 registers) still fits on
the sleeper's CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
cases = ((1024, 64, 0, "1 wave, few registers, 1 KiB: an update workgroup still fits"),
         (21 * 1024, 512, 0, "8 waves, few registers, 21 KiB: an update workgroup still fits"),
         (68 * 1024, 64, 0, "1 wave, few registers, 68 KiB of LDS: no update workgroup fits (LDS)"),
         (21 * 1024, 512, 4 | (2 << 5), "8 waves x 148 registers, 21 KiB: no update workgroup fits (registers)"),
         (21 * 1024, 256, 4 | (2 << 5), "4 waves x 148 registers, 21 KiB: an update workgroup still fits (2 x 144 + 148 <= 512)"),
         (21 * 1024, 512, 4 | (3 << 5), "8 waves x 108 registers, 21 KiB: an update workgroup still fits (2 x 144 + 2 x 108 <= 512)"),
         (150 * 1024, 512, 4 | (2 << 5), "8 waves x 148 registers, 150 KiB: nothing else fits at all"))
for form, pers, tk in (("per-tile", 0, 0), ("persistent + stealing", 1, 0), ("ticket", 0, 2)):
    ctx.set_option("gemm_persist", pers); ctx.set_option("gemm_ticket", tk)
    base = ctx.probe_gemm(M, N, K, 1, 32, 6)
    print("%s alone: %.2f TF/s (%.3f ms per launch)" % (form, base[0], base[1]), flush=True)
    for lds, thr, fl, what in cases:
        t0 = time.perf_counter()
        check(lib.gpmi_probe_resident(ctx._h, 1, lds, thr, 400.0, 0, fl))
        t_call = time.perf_counter() - t0
        time.sleep(0.02)
        r = ctx.probe_gemm(M, N, K, 1, 32, 6)
        print("   beside one sleeping workgroup (%s) [probe call %.0f ms]: %.2f TF/s (%+.1f %%)" % (what, t_call * 1e3, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
        time.sleep(0.3)
