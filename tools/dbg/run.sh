#!/bin/bash
timeout 1200 python -m pytest tests -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tail -8
timeout 600 python tools/bench_binding.py 2>&1 | tail -1 | tee gpurun_out/binding_ab.json
