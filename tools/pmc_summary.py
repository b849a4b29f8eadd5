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

# which blend backward the workload ran: the parity-mode kernel in the strict / exact modes (bench.py --mode), the S = 0 default
# instantiation in the fast mode
MODE = os.environ.get("SGR_PMC_MODE", "strict")
KERNEL = "sgr_blend_bwd_kernel_exact" if MODE in ("strict", "exact") else "sgr_blend_bwd_kernel_s0"


def per_kernel_means(path):
    acc = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def kernel_duration_ns(path):
    """Mean duration of the blend-backward dispatches in a pass's kernel trace (the clock of a profiled pass differs from
    an un-profiled one, so derived rates use the duration measured in the SAME pass)."""
    if not os.path.exists(path):
        return None
    tot, n = 0.0, 0
    try:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if KERNEL in row.get("Kernel_Name", ""):
                    tot += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    n += 1
    except (KeyError, ValueError, OSError):
        return None
    return tot / n if n else None


def derive(res):
    """Counted (not estimated) figures of the dominant kernel.  Units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
    SQ_ACTIVE_INST_* are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_BUSY_CYCLES over the
    32 shader engines; 256 CUs x 4 SIMDs."""
    d = {}
    g = res.get("grbm_gui_active")
    dur = res.get("kernel_ns_under_sq2_pass")
    if g:
        cyc = g / 8.0  # cycles the kernel was resident, per XCD
        d["kernel_cycles"] = cyc
        if dur:
            d["effective_clock_ghz"] = round(cyc / dur, 3)
        if res.get("sq_insts_valu"):
            try:
                vm = json.load(open(os.path.join(ROOT, "profiles", "r6", "valu_model.json")))["parity_mode" if MODE in ("strict", "exact") else "default"]
                if vm.get("source_sha16") == res.get("source_sha16"):
                    d["valu_pipe_cycles_per_visit_model"] = vm["valu_pipe_cycles_per_visit"]
            except (OSError, KeyError, ValueError):
                pass
        if res.get("sq_lds_idx_active"):
            d["lds_array_busy_frac"] = round(res["sq_lds_idx_active"] / 256.0 / cyc, 3)
        if res.get("sq_insts_salu"):
            d["salu_insts_per_cu_cycle"] = round(res["sq_insts_salu"] / 256.0 / cyc, 3)
    if res.get("sq_thread_cycles_valu") and res.get("sq_insts_valu"):
        d["exec_mask_lane_util"] = round(res["sq_thread_cycles_valu"] / res["sq_insts_valu"] / 64.0, 3)
    wc = res.get("sq_wave_cycles")
    if wc:
        for k in ("sq_wait_inst_any", "sq_wait_any", "sq_active_inst_any", "sq_wait_inst_lds", "sq_active_inst_valu",
                  "sq_active_inst_lds", "sq_active_inst_sca"):
            if res.get(k) is not None:
                d[k.replace("sq_", "") + "_over_wave_cycles"] = round(res[k] / wc, 3)
    return d


def main():
    ev = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "pmc_blend_bwd.json")
    res = {"gaussians": int(os.environ.get("SGR_BENCH_P", "1000000")), "width": 1920, "height": 1280, "mode": MODE, "semantics": 0,
           "source_sha16": sgr_build.source_sha16(), "fetch_correction": 2.0,
           "source": f"{ev}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / three SQ(+GRBM) groups, separate passes over "
                     "profiles/pmc_workload.py (tools/gpu_evidence.sh)"}
    for name, sub in (("fetch", "fetch"), ("write", "write"), ("sq", "sq"), ("sq2", "sq2"), ("sq3", "sq3")):
        p = os.path.join(ev, sub, f"{sub}_counter_collection.csv")
        if not os.path.exists(p):
            continue
        means, counts = per_kernel_means(p)
        dur = kernel_duration_ns(os.path.join(ev, sub, f"{sub}_kernel_trace.csv"))
        if dur:
            res[f"kernel_ns_under_{name}_pass"] = dur
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
    res["derived"] = derive(res)
    # hit-lane utilisation of the walk (pixels that pass the alpha test per visit / 64): not a hardware counter -- from the
    # CPU replay of the benchmark scene's lists (tools/sim_tile_order.py: 157 993 941 lane hits over 5 573 172 visits)
    res["derived"]["hit_lane_util_replay"] = round(157993941.0 / 5573172.0 / 64.0, 3)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
