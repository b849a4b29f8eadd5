#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "giant" 2>&1 | grep -v amdgpu.ids | tail -5
