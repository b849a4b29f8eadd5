#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "knn" 2>&1 | grep -v amdgpu.ids | tail -4
python - <<'PY'
import torch, time
from simple_knn._C import distCUDA2
for n in (200000, 1000000):
    g = torch.Generator().manual_seed(7)
    pts = (torch.randn(n, 3, generator=g) * torch.tensor([30.0, 3.0, 40.0])).cuda()
    distCUDA2(pts); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): distCUDA2(pts)
    torch.cuda.synchronize()
    print(n, "distCUDA2 ms", round(1e3 * (time.perf_counter() - t) / 5, 3))
PY
