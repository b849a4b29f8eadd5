#!/usr/bin/env python
"""Which reference cycle keeps ~1.25 GB per iteration alive until Python's cyclic GC runs (tools/densify_mem_trace.py: the
allocated bytes of bench.DensifyLoop climb for ~10 steps, then drop)?  One step under gc.DEBUG_SAVEALL, the garbage listed."""
import os, sys, gc, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from types import SimpleNamespace
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
args = SimpleNamespace(width=1920, height=1280, reduce="factored", exchange="blocking")
dev = torch.device("cuda:0")
GB = 1 << 30
loop = bench.DensifyLoop(args, P, dev, 10)
for _ in range(3):
    loop.step()
torch.cuda.synchronize(); gc.collect()
a0 = torch.cuda.memory_allocated()
gc.disable()
for i in range(4):
    loop.step(); torch.cuda.synchronize()
    print(f"step {i}: allocated {torch.cuda.memory_allocated()/GB:.3f} GB (+{(torch.cuda.memory_allocated()-a0)/GB:.3f})", flush=True)
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print("unreachable objects found:", n, "-> allocated", round(torch.cuda.memory_allocated()/GB, 3))
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(25))
for o in gc.garbage:
    if isinstance(o, torch.Tensor) and o.numel() * o.element_size() > (1 << 24):
        print("tensor", tuple(o.shape), o.dtype, o.numel() * o.element_size() >> 20, "MiB; referrers:",
              [type(r).__name__ for r in gc.get_referrers(o)][:6])
for o in gc.garbage:
    tn = type(o).__name__
    if tn not in ("Tensor", "tuple", "list", "dict", "cell", "function", "method"):
        print("obj", tn, repr(o)[:160])
gc.set_debug(0); gc.garbage.clear(); gc.enable()
