"""How many streams of a priority run side by side on this box?  (gpmi_probe_stream_overlap: one 20-ms sleeping kernel per
stream; wall / 20 ms = how many had to queue behind each other)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_amd import GPContext, _lib
from gaussian_process_amd._lib import check
ctx = GPContext(0)
lib = _lib.load()
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for nh, nn in ((1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (8, 0), (0, 1), (0, 2), (0, 4), (0, 5), (0, 8), (2, 2), (4, 4), (2, 1), (3, 1)):
    w = C.c_double()
    check(lib.gpmi_probe_stream_overlap(ctx._h, nh, nn, 20.0, C.byref(w)))
    print("%d high + %d normal streams, one 20-ms kernel each: wall %.1f ms = %.2f x" % (nh, nn, w.value, w.value / 20.0), flush=True)
