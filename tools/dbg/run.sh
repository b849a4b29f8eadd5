cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q --tb=short -m gpu -x 2>&1 | grep -v amdgpu.ids | grep -v "^{" | grep -E "passed|failed|error|Error|assert" | tail -8
