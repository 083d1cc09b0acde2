"""bench.py --gpus N started WITHOUT a launcher must run N ranks (or refuse), never a silent 1-rank run that prints
``n_gpus: 1``.  No GPU here: the ranks get as far as the device check, which names
their rank and world size -- proof that bench.py re-executed itself under torch.distributed.run with WORLD_SIZE = N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_relaunch_command_runs_one_process_per_gpu():
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.remove(ROOT)
    cmd = bench.relaunch_command(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"


def test_bench_without_a_launcher_starts_n_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check (on a GPU box the ranks would run the whole benchmark)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "launching -m torch.distributed.run" in r.stderr
    # both ranks reached bench.py's own device check with the world size it asked for
    assert "[rank 0 of 2]" in r.stderr and "[rank 1 of 2]" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus"' not in r.stdout


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    assert '"n_gpus"' not in r.stdout
