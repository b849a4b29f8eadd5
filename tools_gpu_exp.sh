#!/bin/bash
run() { timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['stages_ms']['blend_fwd'], 'bwd', d['roofline']['stages_ms']['blend_bwd'], 'gauss', d['roofline']['stages_ms']['gauss_bwd'])"; }
echo "== det"; run
echo "== nodet"; SGR_NO_DET=1 run
echo "== 2M"; timeout 600 python bench.py --no-cpu-baseline --gaussians 2000000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['num_rendered_R'], d['roofline']['stages_ms'])"
echo "== 500k"; timeout 600 python bench.py --no-cpu-baseline --gaussians 500000 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['num_rendered_R'], d['roofline']['stages_ms'])"
echo "== 2M S=19"; timeout 600 python bench.py --no-cpu-baseline --gaussians 2000000 --semantics 19 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['num_rendered_R'], d['roofline']['stages_ms'])"
