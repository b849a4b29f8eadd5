#!/usr/bin/env python
"""Per-kernel means of every counter found in the rocprofv3 `*_counter_collection.csv` files below a directory (one
sub-directory per `--pmc` pass, as tools/gpu_pmc3.sh writes them).

    python tools/pmc_table.py gpurun_out/pmc3_1M [out.json] [kernel-name filter, default sgr_]

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE counts half of a wide read stream (MI355X_MICROARCH.md, HBM;
the 1 GiB calibration copy at the head of profiles/pmc_workload.py shows it), so `hbm_bytes` = 1024 * (2 * FETCH_SIZE +
WRITE_SIZE); both raw figures are kept."""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    filt = sys.argv[3] if len(sys.argv) > 3 else "sgr_"
    acc = {}
    for path in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if filt not in k and "copyBuffer" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {}
    for k, cs in acc.items():
        r = {c: v[0] / v[1] for c, v in cs.items()}
        r["dispatches"] = max(v[1] for v in cs.values())
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["hbm_bytes"] = int(1024 * (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]))
        if r.get("SQ_BUSY_CYCLES") and r.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU: quad-cycles some wave of the SIMD issues VALU, summed over SIMDs; SQ_BUSY_CYCLES:
            # cycles the SQ is busy, per shader engine (32 of them)
            r["note"] = "see tools/pmc_blend.py for the derived VALU-busy figure"
        res[k] = r
    txt = json.dumps(res, indent=1, sort_keys=True)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
