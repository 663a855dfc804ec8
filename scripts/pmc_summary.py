"""Summarise rocprofv3 --pmc counter_collection.csv per kernel: sum and mean per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else None
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0]
        if only and only not in k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-34s sum %.6g   dispatches %d   mean %.6g" % (c, v, cnt[k][c], v / cnt[k][c]))
