"""How many BUSY queues does it take to slow the per-tile update GEMM?  One 32768 x 32768 x 2048 lower update (per-tile form,
six launches) on the context's stream beside: k resident sleepers, each on a high-priority stream of its own (a queue that
always has an unfinished kernel), optionally plus a storm of short kernels on yet another high-priority stream."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
ctx.set_option("gemm_persist", 0)
base = ctx.probe_gemm(M, N, K, 1, 0, 5)
print("GPU_MAX_HW_QUEUES=%s per-tile alone: %.2f TF/s (%.3f ms per launch)" % (os.environ.get("GPU_MAX_HW_QUEUES"), base[0], base[1]), flush=True)
for k, hp, storm in ((1, 1, 0), (2, 1, 0), (3, 1, 0), (1, 0, 0), (2, 0, 0), (1, 1, 1), (2, 1, 1), (0, 1, 1)):
    for _ in range(k):
        check(lib.gpmi_probe_resident(ctx._h, hp, 21 * 1024, 512, 450.0, 2, 2))
    th = None
    if storm:
        th = threading.Thread(target=lambda: check(lib.gpmi_probe_launch_storm(ctx._h, 1, 8000, 30.0, 0)))
        th.start()
    time.sleep(0.02)
    r = ctx.probe_gemm(M, N, K, 1, 0, 5)
    if th:
        th.join()
    print("   beside %d resident workgroup(s) on %s-priority streams of their own%s: %.2f TF/s (%+.1f %%)"
          % (k, "high" if hp else "normal", " + a storm of 30-us kernels on another high-priority stream" if storm else "", r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
    time.sleep(0.7)
