"""Sums rocprofv3 --pmc counter_collection.csv per kernel name: python scripts/pmc_summary.py <csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    n = len(cnt[k])
    print(f"{k[:120]}  ({n} dispatches)")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v / n:16.1f} per dispatch")
