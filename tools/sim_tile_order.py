#!/usr/bin/env python
"""CPU-side study of the blend walks on the benchmark scene (no GPU): replays the oracle's per-tile lists
(tools/replay_visits.c, built into /tmp on the fly), prints the visit statistics quoted in DESIGN.md sections 3 and 10
-- (quadrant, instance) visits, lanes hitting, what 4x4 / 8x2 / 8x4 sub-wave units would need -- and simulates how the
9600 tiles of a launch fill 8 XCDs x 32 CUs x 8 workgroup slots (processor sharing per CU) in the shipped supertile order
and longest-first.  Uses oracle/ as the source of the lists: a measurement tool, not part of the product.

    python tools/sim_tile_order.py [gaussians]
"""
import subprocess
import sys, os, ctypes as C, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np, torch
from helpers import oracle_kwargs
from oracle import oracle
from street_gaussians_amd import synthetic as syn
cam = syn.make_camera(1920, 1280, fx=2050.0)
sc = syn.make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, cam, S=0, seed=0)
fw = oracle.forward(**oracle_kwargs(cam, sc))
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", os.path.join(HERE, "replay_visits.c"), "-o",
                       "/tmp/replay_visits.so", "-lm"])
L = C.CDLL("/tmp/replay_visits.so")
p = lambda a: a.ctypes.data_as(C.c_void_p)
gx, gy = 120, 80
out = np.zeros(16); cost = np.zeros(gx * gy, np.float32)
arrs = [np.ascontiguousarray(x) for x in (fw.ranges.astype(np.uint32), fw.point_list.astype(np.uint32), fw.means2D.astype(np.float32), fw.conic_opacity.astype(np.float32), fw.n_contrib.astype(np.uint32))]
L.replay(1920, 1280, *[p(a) for a in arrs], 128, p(out), p(cost))
names = ["quadrant_visits", "max4_4x4_per_batch", "max4_8x2_per_batch", "max2_8x4_per_batch", "lane_hits", "sum_4x4_visits",
         "max4_4x4_nobatch", "rounds", "max16_per_batch", "sum_8x4_visits", "sum_round_max_quadrant", "sum_round_mean_quadrant",
         "visits_confined_to_one_half", "visits_confined_to_one_row"]
print({n: float(out[i]) for i, n in enumerate(names)})
if os.environ.get("SGR_SIM_STATS_ONLY"):
    sys.exit(0)
# barrier skew of the walk: a round ends when its busiest quadrant is done; the other three waves wait at the barrier
for rb in (64, 128, 256, 512, 1 << 20):
    o2 = np.zeros(16); c2 = np.zeros(gx * gy, np.float32)
    L.replay(1920, 1280, *[p(a) for a in arrs], rb, p(o2), p(c2))
    print(f"round of {rb if rb < 1 << 20 else 'whole list'} entries: rounds {o2[7]:.0f}, busiest-quadrant visits {o2[10]:.0f}, mean-quadrant visits "
          f"{o2[11]:.0f} -> {100 * (o2[10] / o2[11] - 1):.1f} % of a wave's walk time waiting for the busiest quadrant")
print("tile cost: mean", cost.mean(), "max", cost.max(), "min", cost.min(), "p10", np.percentile(cost, 10), "p90", np.percentile(cost, 90))
ST = int(os.environ.get("SGR_SIM_ST", "8"))
sgx = (gx + ST - 1) // ST; sgy = (gy + ST - 1) // ST
nst = sgx * sgy
nblocks = ((nst + 7) // 8) * 8 * ST * ST
def tile_of(b):
    x, q = b & 7, b >> 3
    st, within = (q // 64) * 8 + x, q % 64
    tx, ty = (st % sgx) * ST + within % ST, (st // sgx) * ST + within // ST
    return ty * gx + tx if tx < gx and ty < gy else -1
def simulate(order_per_xcd, slots_per_cu=8, cus=32):
    """processor sharing per CU: each CU has unit capacity shared by its resident WGs; a new WG goes to the CU with a free slot (fewest resident)."""
    worst = 0.0
    for tasks in order_per_xcd:
        tasks = [c for c in tasks]
        # event simulation
        res = [[] for _ in range(cus)]  # remaining work of resident WGs per CU
        t = 0.0; i = 0
        # initial fill round robin
        while i < len(tasks) and any(len(r) < slots_per_cu for r in res):
            k = min(range(cus), key=lambda c: len(res[c])); res[k].append(tasks[i]); i += 1
        while any(res):
            # next completion: per CU, the WG with least remaining finishes after rem * n
            best = None
            for c in range(cus):
                if res[c]:
                    m = min(res[c]); dt = m * len(res[c])
                    if best is None or dt < best[0]: best = (dt, c)
            dt, cfin = best
            for c in range(cus):
                if res[c]:
                    dec = dt / len(res[c])
                    res[c] = [r - dec for r in res[c]]
            t += dt
            for c in range(cus):
                done = [r for r in res[c] if r <= 1e-9]
                if done:
                    res[c] = [r for r in res[c] if r > 1e-9]
                    for _ in done:
                        if i < len(tasks):
                            res[c].append(tasks[i]); i += 1
        worst = max(worst, t)
    return worst
cur = [[] for _ in range(8)]
for b in range(nblocks):
    tl = tile_of(b)
    cur[b & 7].append(float(cost[tl]) if tl >= 0 else 0.0)
ideal = cost.sum() / 256
print("ideal (perfect balance over 256 CUs):", ideal)
print("current order makespan:", simulate(cur))
lpt = [sorted(x, reverse=True) for x in cur]
print("LPT within XCD:", simulate(lpt))
allc = sorted([float(c) for c in cost], reverse=True)
lptg = [allc[x::8] for x in range(8)]
print("global LPT dealt over XCDs:", simulate(lptg))
