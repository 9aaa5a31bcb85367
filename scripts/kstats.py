"""Prints a rocprofv3 *_kernel_stats.csv as a short table (name, calls, average us, share)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) < float(sys.argv[2] if len(sys.argv) > 2 else 0.3):
        continue
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={float(r['Percentage']):.2f}")
