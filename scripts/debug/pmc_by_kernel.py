"""Sums rocprofv3 *_counter_collection.csv per (kernel, grid, counter): mean per dispatch.
`python scripts/debug/pmc_by_kernel.py <counter_collection.csv> [substring]`"""
import collections
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(f"{k[0]:60s} grid={k[1]:>8s} {k[2]:28s} n={len(v):4d} mean={sum(v) / len(v):14.1f}")
