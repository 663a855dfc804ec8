"""Summaries from rocprofv3's rocpd (sqlite) output, the default format of this ROCm.

usage:
  rocpd_extract.py stats   <results.db> <out.csv>          per-kernel calls / total / avg / min / max (ns)
  rocpd_extract.py traffic <fetch.db> <write.db> <out.json> [kernel substring]
       HBM bytes per trailing-update launch from separate FETCH_SIZE / WRITE_SIZE passes
       (KiB; FETCH_SIZE x2 on gfx950 for 16 B/lane reads, see collect_roofline.py)
  rocpd_extract.py pmc     <results.db> <out.txt>           per-kernel sums of every collected counter
"""
import csv
import json
import sqlite3
import sys
from collections import defaultdict


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, k, s, a, lo, hi in rows:
            w.writerow([n, k, s, "%.1f" % a, "%.3f" % (100.0 * s / tot), lo, hi])
    for r in rows[:8]:
        print("%-90s calls %6d total %10.3f ms avg %9.1f us" % (r[0][:90], r[1], r[2] / 1e6, r[3] / 1e3))


def per_dispatch(db, counter, kernel):
    c = sqlite3.connect(db)
    return [(v, d / 1e6) for v, d in c.execute(
        "select value, duration from counters_collection where counter_name=? and kernel_name like ?",
        (counter, "%" + kernel + "%"))]


def traffic(fdb, wdb, out, kernel="chol_trailing_update_dma_kernel"):
    min_ms = 0.0
    f = per_dispatch(fdb, "FETCH_SIZE", kernel)
    w = per_dispatch(wdb, "WRITE_SIZE", kernel)
    fl = [v for v, ms in f if ms >= min_ms]
    wl = [v for v, ms in w if ms >= min_ms]
    fb = 2.0 * 1024.0 * sum(fl) / max(len(fl), 1)
    wb = 1024.0 * sum(wl) / max(len(wl), 1)
    res = {"kernel": kernel, "launches_counted": len(fl), "min_launch_ms": min_ms,
           "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
           "fetch_size_correction": 2.0,
           "total_fetch_bytes": 2.0 * 1024.0 * sum(v for v, _ in f),
           "total_write_bytes": 1024.0 * sum(v for v, _ in w),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0"}
    # the source the counters describe: bench.py marks roofline.traffic "stale" when this file has changed since
    import hashlib, os
    src = os.path.join("gaussian_process_amd", "csrc", "gemm_dma.hip")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        res["kernel_source_sha256"] = hashlib.sha256(open(os.path.join(root, src), "rb").read()).hexdigest()
        res["kernel_source"] = src
    except OSError:
        pass
    res["git"] = os.environ.get("GPMI_GIT_REV")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


def pmc(db, out):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for k, n, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
        acc[k][n] += v
    for k, n in c.execute("select kernel_name, count(distinct dispatch_id) from counters_collection group by kernel_name"):
        cnt[k] = n
    with open(out, "w") as f:
        for k in sorted(acc, key=lambda k: -max(acc[k].values())):
            f.write("%s  (dispatches %d)\n" % (k, cnt[k]))
            for n in sorted(acc[k]):
                f.write("    %-36s %.6g\n" % (n, acc[k][n]))
    print(open(out).read()[:2000])


if __name__ == "__main__":
    {"stats": stats, "traffic": traffic, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
