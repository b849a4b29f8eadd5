cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|assert" | tail -8
b() { timeout 600 python bench.py --no-cpu-baseline --no-other-configs "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stages_ms'])"; }
echo "== W1"; b --steps 300
echo "== W1 again"; b --steps 300
echo "== 2M S19"; b --steps 50 --gaussians 2000000 --semantics 19
echo "== 5M"; b --steps 30 --gaussians 5000000
