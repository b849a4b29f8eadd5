#!/usr/bin/env python
"""Memory trace of bench.DensifyLoop (configs[4], 5 M Gaussians): allocated / reserved bytes at the points of interest and the
largest live tensors' origin, to see what a loop that re-sizes under load really holds.  GPU box only.
    python tools/densify_mem_trace.py [gaussians] [pool_factor]"""
import os, sys, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from types import SimpleNamespace

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
args = SimpleNamespace(width=1920, height=1280, reduce="factored", exchange="blocking")
dev = torch.device("cuda:0")
GB = 1 << 30
say = lambda tag: print(f"{tag:40s} allocated {torch.cuda.memory_allocated()/GB:6.2f} GB  reserved {torch.cuda.memory_reserved()/GB:6.2f} GB  "
                        f"peak_alloc {torch.cuda.max_memory_allocated()/GB:6.2f}", flush=True)
say("start")
loop = bench.DensifyLoop(args, P, dev, 10)
say("loop constructed")
orig_step, orig_dens = loop.step, loop.densify
n = [0]
def step():
    orig_step()
    n[0] += 1
    if n[0] in (1, 2, 10, 11, 12, 20, 21, 30, 31, 40):
        torch.cuda.synchronize(); say(f"after step {n[0]} (P={loop.P})")
def dens():
    say("before densify")
    orig_dens()
    say(f"after densify (P={loop.P})")
loop.step, loop.densify = step, dens
res = loop.run(lambda: torch.cuda.synchronize())
say("after run")
print({k: res[k] for k in ("ms_per_step_amortised", "raster_ms_steady_median", "device_allocations_in_region", "allocator", "pool_bytes_reserved")})
del loop
gc.collect(); torch.cuda.empty_cache(); say("after del + empty_cache")
