"""The resident potrf128 server (never used) beside the update GEMM and NOTHING else: is it the workgroup or the context?"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
for form, pers in (("per-tile", 0), ("persistent + stealing", 1)):
    ctx.set_option("gemm_persist", pers)
    for mode, what in ((0, "no server"), (9, "server resident (8 waves, 139 registers, 7 parked at a barrier)"), (9 + 256, "server, waves 1-7 gone"),
                       (9 + 512, "server, waves 1-7 polling LDS with s_sleep instead of the barrier"), (9 + 16 + 512, "the same, wave 0 polling slowly too"), (0, "no server")):
        out = (C.c_double * 2)()
        check(lib.gpmi_probe_gemm_beside_server(ctx._h, 32768, 32768, 2048, 1, 32, 6, mode, out))
        print("%s, %s: %.2f TF/s (%.3f ms per launch)" % (form, what, out[0], out[1]), flush=True)
