"""Turn rocprofv3 PMC passes over `bench.py --steps 1 --warmup 0` into profiles/<tag>_roofline_traffic.json.

usage: collect_roofline.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel substring]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section), which is the access form of
the operand staging (global_load_lds_dwordx4) that dominates this kernel's reads; the
correction (x2) is applied and recorded.  Per launch = mean over the kernel's dispatches
that ran longer than `min_ms` (the trailing-update launches; the same kernel also serves
short in-panel updates, which are excluded).
"""
import csv
import json
import sys
from collections import defaultdict


def per_dispatch(path, counter, kernel):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                out[int(r["Dispatch_Id"])] = (float(r["Counter_Value"]),
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return out


def main():
    fetch_csv, write_csv, out_json = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "gemm_nt_dma_kernel"
    min_ms = 0.5
    f = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_dispatch(write_csv, "WRITE_SIZE", kernel)
    fl = [v for v, ms in f.values() if ms >= min_ms]
    wl = [v for v, ms in w.values() if ms >= min_ms]
    fetch_b = 2.0 * 1024.0 * sum(fl) / max(len(fl), 1)
    write_b = 1024.0 * sum(wl) / max(len(wl), 1)
    res = {"kernel": kernel, "launches_counted": len(fl), "min_launch_ms": min_ms,
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "traffic_bytes_per_launch": fetch_b + write_b,
           "fetch_size_correction": 2.0,
           "total_fetch_bytes": 2.0 * 1024.0 * sum(v for v, _ in f.values()),
           "total_write_bytes": 1024.0 * sum(v for v, _ in w.values()),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 0"}
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
