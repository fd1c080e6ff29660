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


import pytest


@pytest.mark.gpu
def test_gpu_bench_rccl_path_gives_the_same_records():
    """One rank through the whole multi-GPU path of bench.py (ISAAC_BENCH_FORCE_DIST=1: process group over RCCL, all-reduce of the contig
    flags, broadcast of the template statistics, every step's records and CIGARs gathered behind the later steps) against the plain
    one-GPU run on a small workload: the gathered step is the step that was computed, and both runs produce the same records."""
    args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--pairs-per-step", "60000", "--genome-bases", "30000000", "--no-pcie-pass", "--no-bam-pass", "--cpu-sample-pairs", "20000"]
    lines = []
    for env in ({}, {"ISAAC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}):
        r = run(args + (["--broadcast-index"] if env else []), env=env)          # the RCCL run also takes its table through shard.broadcast_table
        assert r.returncode == 0, r.stderr[-3000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    plain, dist = lines
    assert plain["parity_diffs"] == 0 and dist["parity_diffs"] == 0
    assert "gather_identical" not in plain and dist["gather_identical"] is True
    assert plain["records_sha1"] == dist["records_sha1"]
    assert plain["n_gpus"] == dist["n_gpus"] == 1 and dist["value"] > 0


def _device_count():
    import torch
    return torch.cuda.device_count()           # (counts without initialising the GPU runtime)


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL between two ranks (the one-GPU boxes of the pool run the one-rank form above)")
def test_gpu_bench_two_ranks_rccl():
    """BASELINE configuration 3 on the smallest scale it exists at: two ranks, one per GPU, started as the driver starts them
    (`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`), reads sharded statically, the contig flags all-reduced, the template statistics
    broadcast, the table broadcast over the link, every step's records gathered to rank 0 over RCCL.  Rank 0's first step is the one-GPU run's first step (same
    seed): same records, and the gathered copy is the computed one; the whole-job value covers both ranks."""
    args = ["--steps", "2", "--warmup", "1", "--pairs-per-step", "60000", "--genome-bases", "30000000", "--no-pcie-pass", "--no-bam-pass", "--cpu-sample-pairs", "20000"]
    plain = run(["--gpus", "1"] + args)
    assert plain.returncode == 0, plain.stderr[-3000:]
    plain = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--broadcast-index"] + args, capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    two = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["parity_diffs"] == 0 and two["gather_identical"] is True
    assert two["records_sha1"] == plain["records_sha1"]
    assert two["value"] > 0 and two["scaling"] == "weak"
