"""Multi-process read sharding (isaac_aligner_amd/shard.py) on CPU: world_size 2 over gloo.

Each rank runs the path on its own shard of the clusters -- here with the oracle as the per-rank engine, since there is no GPU
in this container -- and exchanges only what bench.py exchanges between ranks: the contig hit flags (all-reduce), the template
length statistics (broadcast from rank 0) and the records (gather).  Rank 0 then checks the gathered records against one
single-process run over all clusters: identical bytes, i.e. the sharded execution is the unsharded one by construction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, result_path):
    import torch.distributed as dist
    import oracle_lib
    from parity_util import make_inputs
    from isaac_aligner_amd import options, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        contigs, bcl, _ = make_inputs(genome_bases=200000, n_pairs=n_pairs, seed=5, n_contigs=3)
        o = oracle_lib.load()
        ref = o.reference(contigs)
        ref.build_index()
        p = options.default_params(150, 150)
        begin, end = shard.shard_bounds(n_pairs, rank, world)
        mine = np.ascontiguousarray(bcl[begin:end])
        # phase 1 on the shard, then the run-wide state
        matches, hits = ref.find_matches(p, mine, len(mine))
        all_hits = shard.reduce_contig_hits(hits, dist)
        tls = ref.determine_tls(p, mine, matches, all_hits)          # every rank learns from its own first clusters ...
        tls = shard.broadcast_tls(tls, dist)                           # ... and rank 0's statistics win
        # phase 2 on the shard
        rec, cig, _ = ref.select(p, mine, matches, tls, all_hits, n_clusters_hint=len(mine))
        rec_bytes = torch.from_numpy(rec.view(np.uint8).reshape(len(rec), -1).copy())
        gathered = shard.gather_records(rec_bytes, dist, rank, world)
        if rank == 0:
            got = torch.cat(gathered).numpy()
            # the unsharded run: same statistics (rank 0's first tile = the run's first tile), all clusters at once
            fm, fh = ref.find_matches(p, bcl, n_pairs)
            assert (fh == all_hits).all()
            frec, fcig, _ = ref.select(p, bcl, fm, tls, fh, n_clusters_hint=n_pairs)
            want = frec.copy()
            got = got.view(frec.dtype).reshape(-1)
            assert len(got) == len(want)
            # cluster ids restart per shard (each rank numbers its own tile from 0) and CIGAR offsets are per buffer
            names = [f for f in want.dtype.names if f not in ("cluster_id", "cigar_offset", "reserved")]
            same = all((got[f] == want[f]).all() for f in names)
            sizes = [len(g) for g in gathered]
            np.save(result_path, np.array([1 if same else 0] + sizes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [601])   # odd: the shards differ in size, the gather pads
def test_two_rank_sharding_matches_single_process(tmp_path, n_pairs):
    port = _free_port()
    result = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(2, port, n_pairs, result), nprocs=2, join=True)
    r = np.load(result)
    assert r[0] == 1
    assert list(r[1:]) == [2 * 301, 2 * 300]


def test_shard_bounds_cover_everything():
    from isaac_aligner_amd import shard
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in edges) - min(e - b for b, e in edges) <= 1


def _step_gather_worker(rank, world, port, result_path):
    import torch.distributed as dist
    from isaac_aligner_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        g = shard.StepGather(dist, rank, world)
        mine = []
        for step in range(3):
            rng = np.random.default_rng(100 * step + rank)
            n = 50 + 7 * rank + step                                      # shards of different sizes
            rec = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8))
            cig = torch.from_numpy(rng.integers(0, 1 << 20, 2 * n + rank, dtype=np.int64).astype(np.int32))
            if step == 2 and rank == 1:
                cig = cig[:0]                                              # a rank without CIGAR words in a step
            mine.append((rec, cig))
            # as bench.py hands a step over: the pool at its fixed capacity of 4 words per record, its fill level in a tensor, every rank's
            # record count known without asking (here 50 + 7 * rank + step)
            pool = torch.full((4 * n,), -1, dtype=torch.int32)
            pool[:cig.shape[0]] = cig
            g.add(rec, pool, torch.tensor([cig.shape[0]], dtype=torch.int64), record_counts=[50 + 7 * r + step for r in range(world)])
        steps = g.finish()
        if rank == 0:
            assert len(steps) == 3
            for step, (recs, cigs) in enumerate(steps):
                assert len(recs) == world and len(cigs) == world
                for r in range(world):
                    rng = np.random.default_rng(100 * step + r)
                    n = 50 + 7 * r + step
                    want_rec = rng.integers(0, 256, (n, 64), dtype=np.uint8)
                    want_cig = rng.integers(0, 1 << 20, 2 * n + r, dtype=np.int64).astype(np.int32)
                    if step == 2 and r == 1:
                        want_cig = want_cig[:0]
                    assert (recs[r].numpy() == want_rec).all() and (cigs[r].numpy() == want_cig).all()
            open(result_path, "w").write("ok")
        else:
            assert steps is None
    finally:
        dist.destroy_process_group()


def test_step_gather_collects_every_step_on_rank_0(tmp_path):
    """shard.StepGather (what bench.py --gpus N uses to move records and CIGARs while later steps compute): asynchronous gathers of
    padded payloads, sizes differing by rank and step, two gloo ranks"""
    result = str(tmp_path / "result")
    mp.spawn(_step_gather_worker, args=(2, _free_port(), result), nprocs=2, join=True)
    assert open(result).read() == "ok"
    from isaac_aligner_amd import shard
    g = shard.StepGather(None, 0, 1)                                       # single process: nothing to exchange
    r, c = torch.zeros((3, 64), dtype=torch.uint8), torch.zeros(5, dtype=torch.int32)
    g.add(r, c)
    g.add(r, c, torch.tensor([2]))
    (r0, c0), (r1, c1) = g.finish()
    assert r0[0] is r and (c0[0] == c).all() and c1[0].shape[0] == 2



def _broadcast_table_worker(rank, world, port, result_path):
    import torch.distributed as dist
    from isaac_aligner_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(9)
        want = torch.from_numpy(np.stack([np.sort(rng.integers(0, 1 << 62, 100003)), rng.integers(0, 1 << 62, 100003)], axis=1).copy())
        got = shard.broadcast_table(want if rank == 0 else None, dist, rank, chunk=4096)      # 25 pieces
        assert got.dtype == torch.int64 and got.shape == want.shape and (got == want).all()
        if rank == 1:
            open(result_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_table_broadcast_from_rank_0(tmp_path):
    """shard.broadcast_table (bench.py --broadcast-index: one index build, the table sent to the other ranks in pieces)"""
    result = str(tmp_path / "result")
    mp.spawn(_broadcast_table_worker, args=(2, _free_port(), result), nprocs=2, join=True)
    assert open(result).read() == "ok"
    from isaac_aligner_amd import shard
    k = torch.arange(10).view(5, 2)
    assert shard.broadcast_table(k, None, 0) is k
