"""What the KERNEL BOUNDARIES of a busy second stream cost the trailing-update GEMM: gpmi_probe_gemm (one 32768 x 32768 x
2048 lower update, the headline's launch shape; six launches = 0.2 s) alone and while a storm of one-wave kernels runs on
another stream (gpmi_probe_launch_storm): per kernel length, kind (sleep only / + agent-scope fences / + atomic store) and
stream priority.  The runtime brackets every kernel with cache maintenance (an acquire in front, a release behind)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
M = N = 32768
K = 2048
ctx.set_option("gemm_persist", 0)       # the per-tile form, as the update runs under lookahead
base = ctx.probe_gemm(M, N, K, 1, 0, 5)
print("alone: %.2f TF/s (%.3f ms per launch)" % base, flush=True)
for hp, sleep_us, kind in ((1, 1000.0, 0), (1, 200.0, 0), (1, 50.0, 0), (1, 10.0, 0), (1, 0.0, 0), (0, 10.0, 0), (1, 50.0, 1), (1, 10.0, 1),
                           (1, 50.0, 2), (1, 200.0, 1)):
    count = int(min(60000, max(300, 0.35e6 / (sleep_us + 6.0))))          # ~0.35 s of storm
    # the storm is enqueued by a second host thread WHILE the GEMM runs (enqueueing tens of thousands of launches takes
    # as long as executing them: issued first, the storm is over before the GEMM starts)
    import threading
    t_enq = [0.0]

    def storm():
        t0 = time.perf_counter()
        check(lib.gpmi_probe_launch_storm(ctx._h, hp, count, sleep_us, kind))
        t_enq[0] = time.perf_counter() - t0
    th = threading.Thread(target=storm)
    th.start()
    time.sleep(0.01)
    r = ctx.probe_gemm(M, N, K, 1, 0, 5)
    th.join()
    t_enq = t_enq[0]
    ctx.sync() if hasattr(ctx, "sync") else None
    time.sleep(0.6)
    print("beside %5d kernels of %6.1f us (kind %d, priority %s; enqueue took %.0f ms): %.2f TF/s (%+.1f %%)"
          % (count, sleep_us, kind, "high" if hp else "normal", t_enq * 1e3, r[0], 100.0 * (r[0] / base[0] - 1.0)), flush=True)
base2 = ctx.probe_gemm(M, N, K, 1, 0, 5)
print("alone again: %.2f TF/s" % base2[0], flush=True)
