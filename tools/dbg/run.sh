#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_multiview.py -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15
