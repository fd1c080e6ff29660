"""bench.py --gpus N without a launcher starts the ranks itself (a child torch.distributed.run, before any GPU call) and the
world size it reports is the one that ran; with a launcher a mismatch between --gpus and WORLD_SIZE is refused."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)


def test_bench_starts_its_own_ranks():
    r = run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"launch_check": True, "n_gpus": 2, "gpus_requested": 2}


def test_bench_refuses_a_world_size_mismatch():
    r = run(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)
