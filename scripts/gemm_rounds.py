import sys; sys.path.insert(0, "/root/repo")
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for K in (1024, 2048):
    for (M, N) in ((2048, 2048), (4096, 2048), (4096, 4096), (8192, 4096), (8192, 8192), (16384, 8192), (16384, 16384)):
        tf, ms = ctx.probe_gemm(M, N, K, 0, 0, 5)
        tiles = (M // 128) * (N // 128)
        print("K=%d M=%5d N=%5d tiles %5d (%.1f rounds): %.3f ms, %.1f TF/s, %.1f us per round" % (K, M, N, tiles, tiles / 256, ms, tf, ms * 1e3 / (tiles / 256)), flush=True)
