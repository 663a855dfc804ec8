"""Per-stream timeline summary of a rocprofv3 --kernel-trace run (rocpd sqlite): for the LAST `window_ms` of the trace,
per queue and kernel name: calls, summed duration, summed gap to the previous kernel on the same queue.
usage: trace_timeline.py results.db [window_ms]"""
import sqlite3, sys
from collections import defaultdict
db = sys.argv[1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
t_end = max(r[2] for r in rows)
if win:
    rows = [r for r in rows if r[1] >= t_end - win]
t0 = rows[0][1]
print("window: %.3f ms, %d kernels" % ((t_end - t0) / 1e6, len(rows)))
last_end = {}
acc = defaultdict(lambda: [0, 0.0, 0.0])
for name, s, e, q in rows:
    key = (q, name[:70])
    a = acc[key]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if q in last_end:
        a[2] += max(0, s - last_end[q]) / 1e3
    last_end[q] = max(last_end.get(q, 0), e)
busy = defaultdict(float)
for (q, name), (n, d, g) in acc.items():
    busy[q] += d
for q in sorted(busy):
    print("queue %s: busy %.3f ms" % (q, busy[q] / 1e3))
    for (qq, name), (n, d, g) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if qq == q:
            print("   %-70s calls %5d  dur %9.1f us (avg %7.1f)  gap-before %9.1f us (avg %6.1f)" % (name, n, d, d / n, g, g / n))

# raw listing: the kernels that precede the LAST lml_reduce_kernel (end of the last factorisation), all queues
import os
nlist = int(os.environ.get("TRACE_LIST", "0"))
if nlist:
    allrows = c.execute("select name, start, end, %s, grid_x, workgroup_x, lds_size from kernels order by start" % (qcol or "0")).fetchall()
    idx = max(i for i, r in enumerate(allrows) if "lml_reduce" in r[0])
    sel = allrows[max(0, idx - nlist):idx + 1]
    tz = sel[0][1]
    print("---- last %d kernels before the end of the factorisation (us from the first listed start)" % len(sel))
    for name, s, e, q, gx, wx, lds in sel:
        short = name.replace("gpmi::", "").replace("void ", "")[:46]
        print("q%-2s %9.1f -> %9.1f  (%7.1f)  grid %6d x %3d lds %6d  %s" % (q, (s - tz) / 1e3, (e - tz) / 1e3, (e - s) / 1e3, gx // max(wx, 1), wx, lds, short))

# raw listing of the predict phase: everything after the LAST lml_reduce_kernel (TRACE_PREDICT=k kernels)
npred = int(os.environ.get("TRACE_PREDICT", "0"))
if npred:
    allrows = c.execute("select name, start, end, %s, grid_x, workgroup_x, lds_size from kernels order by start" % (qcol or "0")).fetchall()
    idx = max(i for i, r in enumerate(allrows) if "lml_reduce" in r[0])
    sel = allrows[idx:idx + npred]
    tz = sel[0][1]
    print("---- %d kernels from the end of the last factorisation on (predict phase; us)" % len(sel))
    for name, s_, e_, q, gx, wx, lds in sel:
        short = name.replace("gpmi::", "").replace("void ", "")[:46]
        print("q%-2s %9.1f -> %9.1f  (%7.1f)  grid %6d x %3d lds %6d  %s" % (q, (s_ - tz) / 1e3, (e_ - tz) / 1e3, (e_ - s_) / 1e3, gx // max(wx, 1), wx, lds, short))
