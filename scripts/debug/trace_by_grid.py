"""Groups a rocprofv3 *_kernel_trace.csv by (kernel, grid size): calls, median and minimum duration in us.
`python scripts/debug/trace_by_grid.py <kernel_trace.csv> [substring]`"""
import collections
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        d[(r["Kernel_Name"][:70], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(f"{k[0]:70s} grid={k[1]:>8s} n={len(v):4d} med={v[len(v) // 2]:8.1f} min={v[0]:8.1f}")
