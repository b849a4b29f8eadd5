cd $GRAFT_REPO_ROOT
timeout 600 python tools/dbg/prof_scene2.py 2>&1 | grep -v amdgpu.ids | tail -5
