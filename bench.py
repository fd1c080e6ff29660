#!/usr/bin/env python3
"""bench.py -- paired-end reads aligned per second on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a chr21-sized reference (46.7 Mbp; synthetic, because no genome ships with the image),
2x150 bp synthetic pairs, default 10 steps x 1 M pairs = 10 M pairs per GPU.  A step = one pass of the hot path over one
batch of pairs already resident in HBM: isaac_gpu_find_matches (seed extraction + index lookup) followed by
isaac_gpu_select (fragment building, banded Smith-Waterman, mate rescue, MAPQ, clipping, FragmentHeader records).  As in the
reference the two halves run as two phases over all batches, with the loaded-contig set and the template-length statistics
(learnt from the first batch during warm-up, then frozen: MatchSelector.cpp:402-417) fixed in between.
With N > 1 every rank aligns its own K batches on its own GPU (static shard, no data-path collective); the ranks OR-reduce the
per-contig hit flags between the phases and rank 0 gathers the fixed-size alignment records once at the end (RCCL).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs-per-step", type=int, default=1_000_000)
    ap.add_argument("--genome-bases", type=int, default=46_700_000)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--cpu-sample-pairs", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-neighbors", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    os.environ.setdefault("ISAAC_GPU_DEFERRED_COMPLETION", "1")   # read by isaac_gpu_create: select calls pipeline, isaac_gpu_synchronize completes them
    from isaac_aligner_amd import abi, gpu, options, shard, synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("ISAAC_BENCH_FORCE_DIST"):      # the variable: run the collectives with one rank too (1-GPU check of the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = args.read_length
    params = options.default_params(L, L)

    # ---- setup (untimed): reference, index, reads resident in HBM ---------------------------------------------------
    t0 = time.time()
    contigs = synth.make_genome(args.genome_bases, seed=2, device=dev, n_contigs=1)
    al = gpu.Aligner(params, local_rank, contigs)
    n_index = al.build_index(repeat_threshold=1000, annotate_neighbors=not args.no_neighbors)
    t_index = time.time() - t0
    n_batches = args.warmup + args.steps
    batches = []
    for b in range(n_batches):
        bcl, _ = synth.make_read_pairs(contigs, args.pairs_per_step, L, seed=1000 * (rank + 1) + b, device=dev)
        batches.append(bcl)
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    n_rec = args.pairs_per_step * 2
    records = [torch.empty((n_rec, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(args.steps)]
    # only the records are gathered; two CIGAR buffers take turns because the tail of one select call (its wave-per-cluster pass)
    # overlaps the start of the next one (ISAAC_GPU_DEFERRED_COMPLETION, set below)
    cigars = [torch.empty(n_rec * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=dev) for _ in range(2)]

    def reduce_hits(h):
        return shard.reduce_contig_hits(h, dist, dev)

    # ---- warm-up: also learns the template length statistics from the first batch (as tile 1 of the reference does) ---
    tls = None
    for b in range(args.warmup):
        m, o, hits = al.find_matches(batches[b])
        al.set_loaded_contigs(reduce_hits(hits))
        if tls is None:
            tls = al.determine_tls(batches[b], m, o)
        al.select(batches[b], m, o, tls, out=(records[0], cigars[0]))
    if tls is None:
        m, o, hits = al.find_matches(batches[0])
        al.set_loaded_contigs(reduce_hits(hits))
        tls = al.determine_tls(batches[0], m, o)
    shard.broadcast_tls(tls, dist, dev)   # rank 0's statistics are the run's statistics
    al.reset_timers()

    # ---- timed region: exactly K steps ---------------------------------------------------------------------------------
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    found = []
    all_hits = np.zeros(al.n_contigs, np.uint8)
    for s in range(args.steps):                       # phase 1: FindMatchesTransition
        m, o, hits = al.find_matches(batches[args.warmup + s])
        found.append((m, o))
        all_hits |= hits
    al.set_loaded_contigs(reduce_hits(all_hits))      # MatchSelector loads only contigs that received matches
    for s in range(args.steps):                       # phase 2: SelectMatchesTransition
        m, o = found[s]
        al.select(batches[args.warmup + s], m, o, tls, out=(records[s], cigars[s & 1]))
    al.synchronize()                                  # completes the last call's wave-per-cluster pass
    if dist is not None:                              # single gather of the per-GPU records at the end
        shard.gather_records(torch.cat(records), dist, rank, world)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)      # everything resident at the end of the run: index, reads, records, chunk scratch
    counters = al.counters()
    timers = {k: al.kernel_time_ms(k) for k in ("find_matches", "build_fragments", "align_candidates", "finish_candidates", "indel_fragments", "gapped_fragments", "finish_fragments", "plan_rescue", "rescue_windows", "rescue_align",
                                                 "rescue_gapped_plan", "gapped_rescue", "select_order", "select", "select_heavy", "select_residual", "compact_matches")}
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    pairs_total = args.pairs_per_step * args.steps * world
    reads_per_s = 2.0 * pairs_total / elapsed

    # ---- roofline of the dominant kernel: algorithmic bytes (stated per kernel in DESIGN.md) / event-timed duration
    pairs_rank = args.pairs_per_step * args.steps
    log2n = max(1, math.ceil(math.log2(max(2, n_index))))
    seeded_scans = max(0, counters["ungapped_scans"] - counters["rescue_candidates"])
    jobs = counters["rescue_calls"]
    per_kernel_bytes = {
        # BCL in + P * ceil(log2 n) * 16 B index probes + match records out
        "find_matches": 2 * L * pairs_rank + counters["probes"] * log2n * 16 + counters["matches"] * 16,
        # match records in + BCL + one (L + 15)-byte reference window per seeded scan / banded SW + candidate records out
        "build_fragments": counters["matches"] * 16 + 2 * L * pairs_rank + seeded_scans * (L + 15) + counters["candidates"] * 64,
        # per banded Smith-Waterman problem: job record in, read + (L + 15)-base reference window in, result record out
        "gapped_fragments": counters["bsw_jobs"] * (80 + 2 * L + 15 + 232),
        "gapped_rescue": counters["rescue_bsw"] * (80 + 2 * L + 15 + 232),
        # candidate records + gapped results in, consolidated candidate records out
        "finish_fragments": counters["candidates"] * 128 + counters["bsw_jobs"] * 232,
        # one pass over the aligned rescue candidates
        "rescue_gapped_plan": counters["rescue_candidates"] * 64 + jobs * 72,
        # candidate records in, rescue problems out
        "plan_rescue": counters["candidates"] * 64 + jobs * 72,
        # the mate's bases + the window bases in, candidate start positions out
        "rescue_windows": jobs * (72 + L) + counters["rescue_window_bases"] + counters["rescue_candidates"] * 8,
        # per candidate start: the mate (L BCL bytes) + L reference bytes in, one candidate record + 3 cigar words out
        "rescue_align": counters["rescue_candidates"] * (2 * L + 64 + 12),
        # seeded + rescued candidate records in, 2 FragmentHeader records + cigars out
        "select": counters["candidates"] * 64 + counters["rescue_candidates"] * (64 + 12) + 2 * pairs_rank * (64 + 4 * 3),
    }
    total_ms = {k: v[0] * v[1] for k, v in timers.items()}
    heavy_ms = total_ms.pop("select_heavy", 0.0)              # wave-per-cluster pass for clusters that overflow the light work lists
    dominant = max(per_kernel_bytes, key=lambda k: total_ms[k])
    launches = max(1, timers[dominant][1])
    avg_s = total_ms[dominant] / launches / 1e3
    achieved = per_kernel_bytes[dominant] / launches / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure comes from the
    # committed rocprofv3 --pmc passes over this same workload (the latest profiles/*_pmc_summary.json, made by scripts/pmc_traffic.sh + pmc_summary.py;
    # FETCH_SIZE doubled as the MI355X guide prescribes for gfx950), scaled from that run's clusters per launch to this run's
    traffic, traffic_source = None, None
    pmc_files = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_pmc_summary.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    pmc_name = pmc_files[-1] if pmc_files else "r1_k_pmc_summary.json"     # the latest committed passes
    pmc_path = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path)).get("k_" + dominant)
        if pmc and "hbm_bytes_per_launch" in pmc:
            pmc_pairs_per_launch = 500_000.0
            traffic = int(pmc["hbm_bytes_per_launch"] / pmc_pairs_per_launch * (pairs_rank / launches))
            traffic_source = "profiles/%s (separate --pmc FETCH_SIZE / WRITE_SIZE passes, 500000 pairs per launch), scaled per pair" % pmc_name
    roofline = {"bound": "hbm", "kernel": "k_" + dominant, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": round(total_ms[dominant] / launches, 4), "launches": int(launches),
                "algorithmic_bytes_per_launch": int(per_kernel_bytes[dominant] / launches),
                "kernel_ms_total": dict({k: round(v, 2) for k, v in total_ms.items()}, select_heavy=round(heavy_ms, 2)),
                "bytes_per_pair": round(sum(per_kernel_bytes.values()) / pairs_rank, 1),
                # the whole path of this GPU against the same peak (pairs/s x algorithmic bytes per pair)
                "path_achieved": round(sum(per_kernel_bytes.values()) / elapsed / 1e9, 2),
                "path_frac": round(sum(per_kernel_bytes.values()) / elapsed / 1e9 / 8000.0, 6),
                "heavy_clusters": int(counters.get("heavy_clusters", 0))}

    # ---- CPU baseline: the oracle (a port of the reference path) on a bounded sample of the same workload, host cores -------
    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        o = oracle_lib.load()
        cores = os.cpu_count() or 1
        sample = min(args.cpu_sample_pairs, args.pairs_per_step)
        host_bcl = batches[args.warmup][:sample].cpu().numpy()
        ref = o.reference([bytes(c.cpu().numpy()) for c in contigs])
        ref.set_index(al.get_index())
        p = o.default_params(2, L, L)
        find_threads = min(cores, 32)      # every thread streams the whole index for its clusters: more threads only add memory traffic
        tc = time.perf_counter()
        om, ohits = ref.find_matches(p, host_bcl, sample, n_threads=find_threads)
        t_find = time.perf_counter() - tc
        otls = oracle_lib.Tls()
        for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
            setattr(otls, name, getattr(tls, name))
        otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
        tc = time.perf_counter()
        ref.select(p, host_bcl, om, otls, ohits, n_threads=cores, n_clusters_hint=sample)
        t_select = time.perf_counter() - tc
        cpu = {"value": round(2.0 * sample / (t_find + t_select), 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": "%d pairs of the same workload; oracle/ (CPU restatement of the reference path): merge-join seed lookup on %d threads "
                         "(%.2f s) + match selection on %d threads (%.2f s)" % (sample, find_threads, t_find, cores, t_select)}

    out = {"metric": "paired-end reads aligned/sec (2x%dbp)" % L, "value": round(reads_per_s, 1), "unit": "reads/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/int16 (+f64 log-probabilities)", "data": "synthetic",
           "config": {"workload": "chr21-sized synthetic reference (%d bp, 32-mer index %d entries), %d synthetic 2x%d bp pairs per GPU "
                                  "(%d steps x %d pairs)" % (args.genome_bases, n_index, pairs_rank, L, args.steps, args.pairs_per_step),
                      "pairs_per_step": args.pairs_per_step, "read_length": L, "genome_bases": args.genome_bases, "index_entries": int(n_index),
                      "parallelism": "read shards x%d, records gathered once" % world, "hbm_used_gb": hbm_used_gb, "setup_s": round(t_setup, 1), "index_build_s": round(t_index, 1),
                      "tls": list(tls.astuple())},
           "roofline": roofline, "cpu_baseline": cpu,
           "counters": {k: int(v) for k, v in counters.items()}}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
