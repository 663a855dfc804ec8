"""Per-launch durations of the Cholesky's trailing updates in a rocprofv3 --kernel-trace database, in launch order, for the
LAST factorisation in the trace: python scripts/trail_durations.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
tr = [(s, e) for n, s, e in rows if "chol_trailing_update" in n]
lml = [s for n, s, e in rows if "lml_reduce" in n]
last_end = lml[-1]
prev_end = lml[-2] if len(lml) > 1 else 0
sel = [(s, e) for s, e in tr if prev_end < s < last_end]
print("launches", len(sel))
print(" ".join("%.2f" % ((e - s) / 1e6) for s, e in sel))
print("sum %.1f ms" % (sum(e - s for s, e in sel) / 1e6))
srv = [(n, s, e) for n, s, e in rows if "server" in n or "post" in n]
print("server/post kernels:", len(srv), "; longest %.1f ms" % (max((e - s) for n, s, e in srv) / 1e6) if srv else "")
