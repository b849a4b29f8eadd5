"""Prints (kernel, calls, average us) from a rocprofv3 *_kernel_stats.csv for kernels matching a substring."""
import csv
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "sgr_"
for r in csv.DictReader(open(sys.argv[1])):
    if pat in r["Name"]:
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1000:9.1f}")
