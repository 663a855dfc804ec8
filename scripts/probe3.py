import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext
ctx = GPContext(0)
for rule in (0, 1, 2):
    ctx.set_option("gemm_stagger_rule", rule)
    for st in (0, 64):
        ctx.set_option("gemm_stagger", st)
        for (M, N, K, lower) in ((16384, 16384, 512, 0), (16384, 16384, 1024, 0)):
            tf, ms = ctx.probe_gemm(M, N, K, lower, 0, 5)
            print("rule=%d stagger=%d gemm M=%d N=%d K=%d: %.1f TF/s  %.3f ms" % (rule, st, M, N, K, tf, ms), flush=True)
ctx.set_option("gemm_stagger", 0)
for v in (4, 8, 10, 11):
    tf, ms = ctx.probe_gemm(16384, 16384, 512, 0, v, 5)
    print("(dbg build) variant=%2d: %.1f TF/s" % (v, tf), flush=True)
