cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_dbg.json
python -c "
import json; d=json.load(open('gpurun_out/bench_dbg.json')); print(d['value'], d['ms_per_step'], d['sustained'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_source']); print(d['roofline']['stages_ms'])"
timeout 300 python -m pytest tests/test_loss_cpu.py tests/test_gpu_multiview.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
