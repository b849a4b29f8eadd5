"""bench.py's rank launch logic (no GPU): `python bench.py --gpus N` without a launcher must start N ranks itself, refuse
(rc != 0) when fewer than N GPUs exist or when the launcher's world size contradicts --gpus, and run in-process when it
already is a rank."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    assert bench.launch_plan(1, {}, 1, ["--gpus", "1"]) is None
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, 1, []) is None


def test_under_a_launcher_the_process_is_a_rank():
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, 8, []) is None
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(8, {"WORLD_SIZE": "4"}, 8, [])
    assert e.value.code not in (0, None) and "WORLD_SIZE=4" in str(e.value.code)


def test_without_a_launcher_the_ranks_are_spawned():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.launch_plan(8, {}, 8, argv, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    port = int(bench.launch_plan(2, {}, 4, [])[bench.launch_plan(2, {}, 4, []).index("--master-port") + 1])
    assert 1024 <= port < 65536  # a free port was picked


@pytest.mark.parametrize("n,have", [(8, 1), (2, 0), (0, 4)])
def test_fewer_devices_than_ranks_is_an_error(n, have):
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(n, {}, have, [])
    assert e.value.code not in (0, None)


def test_the_script_itself_exits_nonzero_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: no device visible
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to run on fewer ranks" in r.stderr and "{" not in r.stdout
