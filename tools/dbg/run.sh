#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_densify.py tests/test_gpu_loss.py -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -30
timeout 300 python bench.py --scene tests/golden/scene_ref_layout.ply --steps 20 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | cut -c1-600
