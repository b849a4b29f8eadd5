#!/usr/bin/env python
"""Turns the rocprofv3 PMC passes of tools/gpu_evidence.sh (separate --pmc FETCH_SIZE / WRITE_SIZE / SQ passes over
profiles/pmc_workload.py) into profiles/pmc_blend_bwd.json, the file bench.py reads `roofline.traffic` from.

    python tools/pmc_summary.py gpurun_out/ev_<tag> [out.json]

The JSON carries the SHA-256 of the kernel sources it was measured on (street_gaussians_amd/build.py:source_sha16):
bench.py prints the traffic only when that matches the build it is running.
Counter handling (MI355X_MICROARCH.md, HBM / rocprofv3 section; calibration in profiles/README.md): FETCH_SIZE and
WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of a wide read stream (the 1 GiB calibration copy in the
workload shows it), so it is doubled; WRITE_SIZE is used as is."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from street_gaussians_amd import build as sgr_build  # noqa: E402

KERNEL = "sgr_blend_bwd_kernel"


def per_kernel_means(path):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    ev = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_blend_bwd.json")
    res = {"gaussians": int(os.environ.get("SGR_BENCH_P", "1000000")), "width": 1920, "height": 1280,
           "source_sha16": sgr_build.source_sha16(), "fetch_correction": 2.0,
           "source": f"{ev}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ group, separate passes over "
                     "profiles/pmc_workload.py (tools/gpu_evidence.sh)"}
    for name, sub in (("fetch", "fetch"), ("write", "write"), ("sq", "sq")):
        p = os.path.join(ev, sub, f"{sub}_counter_collection.csv")
        if not os.path.exists(p):
            continue
        means, counts = per_kernel_means(p)
        for (kname, counter), v in means.items():
            if KERNEL not in kname:
                continue
            res["kernel"] = kname.split("(")[0]
            res["launches_averaged"] = counts[(kname, counter)]
            if counter == "FETCH_SIZE":
                res["fetch_size_kb_raw"] = v
            elif counter == "WRITE_SIZE":
                res["write_size_kb_raw"] = v
            else:
                res[counter.lower()] = v
        if name == "fetch":  # calibration copy: FETCH_SIZE of a known 256 MiB read
            cal = [v for (kname, counter), v in means.items() if "copyBuffer" in kname and counter == "FETCH_SIZE"]
            if cal:
                res["calibration_copy_fetch_kb"] = cal[0]
    if "fetch_size_kb_raw" in res and "write_size_kb_raw" in res:
        res["hbm_bytes_per_launch"] = int(1024 * (res["fetch_correction"] * res["fetch_size_kb_raw"] + res["write_size_kb_raw"]))
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
