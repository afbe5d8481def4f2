"""`python bench.py --gpus N` must start N ranks itself (or fail loudly), never report a 1-rank run under the label N
(round-1 defect: --gpus was parsed and never read)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_starts_world_of_two(capfd):
    rc = _bench().launch_ranks(2, [], script=os.path.join(ROOT, "tests", "_rank_echo.py"), check_devices=False)
    assert rc == 0
    lines = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                     # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["sum"] == 2 and rec["master"] == "127.0.0.1"


def test_more_ranks_than_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "only" in p.stderr and "GPU" in p.stderr
    assert '"n_gpus"' not in p.stdout


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
