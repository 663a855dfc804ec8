import sys; sys.path.insert(0, "/root/repo")
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for (M, N, K) in ((16256, 128, 128), (16384, 128, 128), (24576, 128, 128), (32640, 128, 128), (32768, 128, 128),
                  (8192, 256, 256), (12288, 256, 256), (16384, 256, 256), (6144, 512, 512), (8064, 512, 512), (8192, 512, 512),
                  (3072, 1024, 1024), (4096, 1024, 1024), (16384, 128, 512), (24576, 128, 1024)):
    tf, ms = ctx.probe_gemm(M, N, K, 0, 0, 20)
    print("M=%6d N=%5d K=%5d tiles128=%4d: %.1f us  %.1f TF/s" % (M, N, K, (M // 128) * ((N + 127) // 128), ms * 1e3, tf), flush=True)
