#!/usr/bin/env python
"""VALU issue model of the blend backward (no GPU): compiles the kernel to ISA, takes the pair loop of the walk (the last
innermost loop of the kernel: two visits per trip), classifies its instructions and prices them with the issue costs
MEASURED on MI355X by tools/ubench/valu_rates.hip (profiles/r4/valu_rates.jsonl: SIMD cycles per wave instruction at eight
waves per SIMD -- plain f32 VALU 2.5-2.7, v_add_f32_dpp 4.4, v_permlane32/16_swap and the transcendentals 8.6, SALU 4.2).
The result -- VALU-pipe cycles per (quadrant, instance) visit -- times the visits of a frame, over the 1024 SIMDs of the chip
at the clock the counters show, is the time the kernel needs if its VALU pipes never idle: the bound bench.py reports as
`roofline.valu_issue`.

    python tools/valu_model.py [kernel name substring]   ->  profiles/r4/valu_model.json
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from street_gaussians_amd import build as b  # noqa: E402

RATES = os.path.join(ROOT, "profiles", "r4", "valu_rates.jsonl")
OUT = os.path.join(ROOT, "profiles", "r4", "valu_model.json")


def measured_costs():
    c = {}
    for ln in open(RATES):
        r = json.loads(ln)
        if r["waves_per_simd"] == 8:
            c[r["inst"]] = r["simd_cycles_per_wave_inst_at_2.4GHz"]
    return {"valu": 0.5 * (c["v_fma_f32"] + c["v_mul_f32"]), "dpp": c["v_add_f32_dpp row_ror"], "swap": c["v_permlane32_swap"],
            "trans": 0.5 * (c["v_exp_f32"] + c["v_rcp_f32"]), "cmp": 0.5 * (c["v_fma_f32"] + c["v_mul_f32"]),
            "salu": c["s_add_u32"], "snop": c["s_nop 0"]}


def classify(line):
    op = line.split()[0]
    if op.startswith("v_permlane"):
        return "swap"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "trans"
    if op.startswith("v_") and "dpp" in line:
        return "dpp"
    if op.startswith("v_cmp"):
        return "cmp"  # (the ubench's 4.4 is a chain on VCC; the kernel's compares write different SGPR pairs)
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_nop"):
        return "snop"
    if op.startswith(("s_waitcnt", "s_barrier")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def model(src, kernel_substr, visits_per_trip=2):
    asm = os.path.join("/tmp", "valu_model_" + src.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc"] + b.flags_for(src) + ["--offload-device-only", "-S", os.path.join(b.CSRC, src), "-o", asm],
                   capture_output=True, text=True, check=True)
    text = open(asm).read()
    names = re.findall(r"^(_Z\S+):", text, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    pick = [n for n, d in zip(names, dem) if kernel_substr in d]
    assert pick, f"no kernel matching {kernel_substr!r}"
    body = text[text.index(pick[0] + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = body.split("\n")
    heads = [i for i, ln in enumerate(lines) if "Inner Loop Header" in ln]
    loop = lines[heads[-1]:]
    counts = collections.Counter()
    for ln in loop:
        ln = ln.strip()
        if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
            continue
        counts[classify(ln)] += 1
    cost = measured_costs()
    valu_cycles = sum(cost[k] * counts[k] for k in ("valu", "dpp", "swap", "trans", "cmp"))
    scalar_cycles = sum(cost["salu"] * counts[k] for k in ("salu", "smem")) + cost["snop"] * counts["snop"]
    return {"kernel": dem[names.index(pick[0])][:90], "source": src, "source_sha16": b.source_sha16(),
            "instructions_per_trip": dict(counts), "visits_per_trip": visits_per_trip,
            "measured_cost_cycles_at_2.4GHz": {k: round(v, 2) for k, v in cost.items()},
            "valu_pipe_cycles_per_visit": round(valu_cycles / visits_per_trip, 1),
            "scalar_pipe_cycles_per_visit": round(scalar_cycles / visits_per_trip, 1),
            "note": "cycles as counted by the microbenchmark at its nominal 2.4 GHz; time bound = visits * cycles / (1024 SIMDs * 2.4e9)"}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "sgr_blend_bwd_kernel_s0<true, true, true>"
    res = {"default": model("sgr_blend_bwd.hip", what), "parity_mode": model("sgr_blend_bwd.hip", "sgr_blend_bwd_kernel_exact<0>"),
           "scalar_walk": model("sgr_blend_bwd_sw.hip", "sgr_blend_bwd_sw_kernel<false>")}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(res, open(OUT, "w"), indent=1)
    for k, v in res.items():
        print(k, v["instructions_per_trip"], "VALU-pipe", v["valu_pipe_cycles_per_visit"], "scalar", v["scalar_pipe_cycles_per_visit"])
