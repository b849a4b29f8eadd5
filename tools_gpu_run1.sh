#!/bin/bash
# first GPU bring-up: primitives -> parity -> golden -> bench
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== primitives"; timeout 900 python -m pytest tests/test_gpu_primitives.py -q --tb=short -m gpu 2>&1 | tail -60 | tee gpurun_out/t_prim.log
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py -q --tb=short -m gpu -k "not full_size" 2>&1 | tail -150 | tee gpurun_out/t_parity.log
echo "== golden"; timeout 600 python tests/golden/make_golden.py gpurun_out/golden 2>&1 | tail -20 | tee gpurun_out/golden.log
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -20 | tee gpurun_out/bench.log
