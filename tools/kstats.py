"""Prints (kernel, calls, average us, total ms per step) from a rocprofv3 *_kernel_stats.csv for kernels matching a substring."""
import csv
import re
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "sgr_"
rows = [r for r in csv.DictReader(open(sys.argv[1])) if pat in r["Name"]]
steps = max([int(r["Calls"]) for r in rows if "blend_fwd" in r["Name"]] or [1])
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    print(f"{name[:58]:58s} calls/step={int(r['Calls']) / steps:5.1f} avg_us={float(r['AverageNs']) / 1000:8.1f} "
          f"ms/step={float(r['TotalDurationNs']) / steps / 1e6:7.4f}")
