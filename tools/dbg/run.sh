#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_loss.py -q --tb=short -m gpu 2>&1 | grep -v amdgpu.ids | tail -8
