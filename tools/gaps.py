#!/usr/bin/env python
"""Reads rocprofv3 kernel traces of bench runs (tools/gpu_gaps.sh) and prints, for the plain training steps in them
(camera pack ... per-Gaussian backward, nothing but the rasterizer's kernels in between), the median wall time of a step,
the kernel time in it, and where the largest gaps between consecutive kernels sit: the time the GPU had nothing queued.

    python tools/gaps.py gpurun_out/gaps_spin gpurun_out/gaps_event
"""
import collections
import csv
import glob
import statistics
import sys

for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print(d, "no kernel trace")
        continue
    rows = list(csv.DictReader(open(f[0])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    starts = [i for i, e in enumerate(ev) if "sgr_pack_camera" in e[2]]  # a step starts with the camera-pack launch
    walls, busy, gaps = [], [], collections.defaultdict(list)
    for a, b in zip(starts[:-1], starts[1:]):
        seg = ev[a:b + 1]
        if not any("blend_bwd" in e[2] for e in seg) or any("at::native" in e[2] for e in seg):
            continue  # forward-only passes, region boundaries (torch kernels of the bench itself)
        walls.append(seg[-1][0] - seg[0][0])
        busy.append(sum(e[1] - e[0] for e in seg[:-1]))
        for i in range(len(seg) - 1):
            gaps[(i, seg[i][2][:32], seg[i + 1][2][:32])].append(seg[i + 1][0] - seg[i][1])
    if not walls:
        print(d, "no steps found")
        continue
    w, k = statistics.median(walls) / 1e3, statistics.median(busy) / 1e3
    print(f"{d}: {len(walls)} steps, median wall {w:.1f} us, kernels {k:.1f} us, idle {w - k:.1f} us per step")
    top = sorted(((statistics.median(v) / 1e3, key) for key, v in gaps.items() if len(v) > len(walls) // 2), reverse=True)[:3]
    for g, (i, p, n) in top:
        print(f"   {g:6.1f} us between launch {i} ({p}) and {i + 1} ({n})")
