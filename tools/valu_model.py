#!/usr/bin/env python
"""VALU issue model of the blend backward (no GPU), round 5.

The kernel's time when its VALU pipes never idle = sum over the frame's visits of the pipe cycles their instruction path
costs, over the 1024 SIMDs of the chip at the measured clock.  Two paths since round 5 (csrc/sgr_blend_bwd.hip):

  dense   per-pixel terms + the 64-lane reduce-scatter (9 permlane swaps, 9 adds, 7 DPP adds) + one store;
  sparse  (k hitting lanes, k <= SGR_SPARSE_K) per-pixel terms + 2 v_mbcnt + address + k - 1 plain adds.

Instruction counts come from the ISA hipcc emits: the pair loop (two visits per trip) of a dense-only build
(-DSGR_SPARSE_K=0) gives the dense path and, split at the first permlane swap of each visit, the part both paths share;
the sparse path's own instructions are read off the shipped build (from the v_mbcnt pair to the `sparse visit done` marker of
the k = 1 case, + k - 1 adds).  Costs are MEASURED IN CYCLES by tools/ubench/valu_rates2.hip (profiles/r5/valu_rates2.jsonl:
launch time x the clock the waves read from s_memtime / wall_clock64, 8 waves per SIMD, >= 20 ms per launch).  The mix of
the paths is the replay's histogram of hitting lanes per visit (tools/lane_hist.py -> profiles/r5/lane_hist_1M.json).

    python tools/valu_model.py   ->  profiles/r6/valu_model.json   (what bench.py reports as roofline.valu_issue)
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

RATES = os.path.join(ROOT, "profiles", "r5", "valu_rates2.jsonl")
HIST = os.path.join(ROOT, "profiles", "r5", "lane_hist_1M.json")
OUT = os.path.join(ROOT, "profiles", "r6", "valu_model.json")
SPARSE_K = 9  # SGR_SPARSE_K of the shipped build


def measured_costs():
    c, clk = {}, []
    for ln in open(RATES):
        r = json.loads(ln)
        if "inst" in r and r["waves_per_simd"] == 8 and r["lanes"] == 64:
            key = [k for k in r if k.endswith("_from_wall_time")]
            c[r["inst"]] = r[key[0]] if key else None
            clk.append(r["clock_ghz"])
    plain = (c["v_fma_f32"] + c["v_mul_f32"] + c["v_add_f32"]) / 3.0
    clk.sort()
    return {"valu": plain, "dpp": c["v_add_f32_dpp row_ror"], "swap": c["v_permlane32_swap"],
            "trans": 0.5 * (c["v_exp_f32"] + c["v_rcp_f32"]), "cmp": c["v_cmp_lt_f32"], "mbcnt": c["v_mbcnt_lo_u32_b32"],
            "pk": c["v_pk_mul_f32"], "salu": c["s_add_u32"], "snop": c["s_nop 0"]}, clk[len(clk) // 2]


def classify(line):
    op = line.split()[0]
    if op.startswith("v_permlane"):
        return "swap"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")):
        return "trans"
    if op.startswith("v_") and "dpp" in line:
        return "dpp"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_mbcnt"):
        return "mbcnt"
    if op.startswith("v_pk_"):
        return "pk"
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


def isa(src, kernel_substr, extra):
    asm = os.path.join("/tmp", "valu_model_" + src.replace(".hip", "") + ("_x" if extra else "") + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + b.flags_for(src) + extra + ["--offload-device-only", "-S", os.path.join(b.CSRC, src), "-o", asm],
                   capture_output=True, text=True, check=True)
    text = open(asm).read()
    names = re.findall(r"^(_Z\S+):", text, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    pick = [n for n, d in zip(names, dem) if kernel_substr in d]
    assert pick, f"no kernel matching {kernel_substr!r}"
    body = text[text.index(pick[0] + ":"):]
    return body[:body.index("s_endpgm")].split("\n"), dem[names.index(pick[0])][:90]


def insts(lines):
    out = []
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
            continue
        out.append(ln)
    return out


def price(counts, cost):
    return sum(cost[k] * n for k, n in counts.items() if k in ("valu", "dpp", "swap", "trans", "cmp", "mbcnt", "pk"))


def model(kernel_substr, extra=()):
    """extra: compiler switches for both ISA dumps (the parity-mode kernel is counted without the rare exact-expf branch of its
    guard, -DSGR_EXACT_BWD_GUARD=0: a static count cannot weigh a branch taken once in 10^4 visits; the guard's own three
    instructions stay in the loop)."""
    cost, clock = measured_costs()
    extra = list(extra)
    # dense-only build: the pair loop = the last innermost loop of the kernel
    lines, name = isa("sgr_blend_bwd.hip", kernel_substr, ["-DSGR_SPARSE_K=0"] + extra)
    # basic blocks carry "in Loop: Header=BBx_y" in their label comments (hipcc places a loop's blocks anywhere in the
    # function): the pair loop = the innermost loop whose blocks hold the twelve permlane32 swaps of two visits
    blocks, cur = {}, None
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)(.*)$", ln)
        if m:
            hdr = re.search(r"Header=BB(\d+_\d+) Depth=(\d+)", m.group(2))
            own = re.match(r"^\.LBB(\d+_\d+):", ln)
            cur = (hdr.group(1), int(hdr.group(2))) if hdr else ((own.group(1), 0) if own else None)
            continue
        if "Loop Header: Depth=" in ln and cur is not None:  # the header block belongs to the loop it heads
            d = int(re.search(r"Depth=(\d+)", ln).group(1))
            cur = (cur[0], d)
            continue
        if cur is not None:
            blocks.setdefault(cur, []).append(ln)
    loop = []
    for key, body in blocks.items():
        cand = insts(body)
        if sum(x.startswith("v_permlane32_swap") for x in cand) == 12 and (not loop or len(cand) < len(loop)):
            loop = cand
    assert loop, "pair loop (two visits of six permlane32 swaps) not found"
    dense_trip = collections.Counter(classify(ln) for ln in loop)
    # the reduce-scatter of one visit: from its first permlane swap up to and including its last DPP add
    first = next(i for i, ln in enumerate(loop) if ln.startswith("v_permlane32_swap"))
    last = first
    for i in range(first, len(loop)):
        if classify(loop[i]) in ("swap", "dpp"):
            last = i
        if loop[i].startswith("ds_write_b32") and i > first + 10:
            break
    reduce = collections.Counter(classify(ln) for ln in loop[first:last + 1])
    # sparse path's own instructions, shipped build: v_mbcnt_lo ... marker of the k = 1 case
    slines, _ = isa("sgr_blend_bwd.hip", kernel_substr, extra)
    sl = insts(slines)
    z = max(i for i, ln in enumerate(sl) if ln.startswith("ds_write_b32") and "offset:44" in ln)  # last store of a stage entry
    a = max(i for i, ln in enumerate(sl[:z]) if ln.startswith("v_mbcnt_lo"))
    assert z - a < 24, "stage stores not found next to their v_mbcnt pair"
    stage = collections.Counter(classify(ln) for ln in sl[a:z])
    stage["valu"] += 2  # the row owners' address add + the zero of the padding word
    dense = price(dense_trip, cost) / 2.0
    red = price(reduce, cost)
    hist = json.load(open(HIST))
    h = hist["hist_hit_lanes"]
    tot = sum(h[1:])
    cyc = 0.0
    sparse_share = 0.0
    for k in range(1, 65):
        if k <= SPARSE_K:
            c = dense - red + price(stage, cost) + (k - 1) * cost["valu"]
            sparse_share += h[k] / tot
        else:
            c = dense
        cyc += h[k] / tot * c
    return {"kernel": name, "source": "sgr_blend_bwd.hip", "source_sha16": b.source_sha16(),
            "instructions_per_trip_dense": dict(dense_trip), "visits_per_trip": 2,
            "reduce_scatter_instructions_per_visit": dict(reduce), "sparse_stage_instructions_per_visit": dict(stage),
            "cost_cycles_per_wave_instruction": {k: round(v, 2) for k, v in cost.items()}, "clock_ghz": round(clock, 3),
            "dense_pipe_cycles_per_visit": round(dense, 1), "reduce_scatter_cycles_per_visit": round(red, 1),
            "sparse_pipe_cycles_per_visit_at_k": {str(k): round(dense - red + price(stage, cost) + (k - 1) * cost["valu"], 1) for k in (1, 4, 9)},
            "sparse_visit_share": round(sparse_share, 4), "valu_pipe_cycles_per_visit": round(cyc, 1),
            "note": "cycles from tools/ubench/valu_rates2.hip (launch time x measured clock / wave-instructions, 8 waves per SIMD); "
                    "mix of the two paths from the replay's hit-lane histogram with the threshold at SGR_SPARSE_K; time bound = visits x cycles / (1024 SIMDs x clock)"}


if __name__ == "__main__":
    res = {"default": model("sgr_blend_bwd_kernel_s0<true, true, true>"), "parity_mode": model("sgr_blend_bwd_kernel_exact<0>", ["-DSGR_EXACT_BWD_GUARD=0"])}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(res, open(OUT, "w"), indent=1)
    for k, v in res.items():
        print(k, "dense", v["dense_pipe_cycles_per_visit"], "reduce", v["reduce_scatter_cycles_per_visit"], "sparse@k",
              v["sparse_pipe_cycles_per_visit_at_k"], "share", v["sparse_visit_share"], "=> per visit", v["valu_pipe_cycles_per_visit"],
              "clock", v["clock_ghz"])
