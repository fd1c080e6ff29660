#!/usr/bin/env python3
"""bench.py -- paired-end reads aligned per second on MI355X (BASELINE.json metric: 2x150 bp, GRCh38).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]): a GRCh38-sized reference -- 3.1 Gbp in 25 contigs with a human-like repeat spectrum,
synthetic because no genome ships with the image (isaac_aligner_amd/synth.py) -- indexed on the device in the sorted-reference
format (2.9 G 32-mer entries, neighbour-annotated), and synthetic 2x150 bp pairs, default 10 steps x 1 M pairs per GPU.
A step = one pass of the hot path over one batch of pairs already resident in HBM: isaac_gpu_find_matches (seed extraction +
index lookup) followed by isaac_gpu_select (fragment building, banded Smith-Waterman, mate rescue, MAPQ, clipping,
FragmentHeader records) and isaac_gpu_compact_cigars.  As in the reference the two halves run as two phases over all batches,
with the loaded-contig set and the template-length statistics (learnt from the first batch during warm-up, then frozen:
MatchSelector.cpp:402-417) fixed in between.

N > 1: one process per GPU.  Started without a launcher (WORLD_SIZE unset), this script starts `torch.distributed.run` with N
ranks itself -- as a child process, before anything touches a GPU -- and exits with its code.  Every rank builds the index on
its own GPU and aligns its own shard of the read batches (no collective on the data path); the ranks OR-reduce the per-contig
hit flags between the phases and rank 0 gathers the alignment records and CIGARs once at the end (RCCL).
--scaling weak (default): every rank aligns K batches of its own; strong: one read set of K batches cut N ways.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--finders", type=int, default=1, help="contexts that do the lookups of the timed region side by side, a host thread each (a lookup ends with one host wait for its match count and contig flags: a single context leaves the GPU idle for it, 0.09 ms a step; with two or three the lookup phase is that much shorter and the step is not: profiles/exp_r6_finders.log)")
    ap.add_argument("--pairs-per-step", type=int, default=1_000_000)
    ap.add_argument("--genome-bases", type=int, default=3_100_000_000)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--indel-read-fraction", type=float, default=None, help="fraction of reads with one indel (default 0.03; config 4: 0.05)")
    ap.add_argument("--indel-max", type=int, default=None, help="longest simulated indel (default 5; config 4: 10)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cpu-sample-pairs", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="also skips the parity check, which uses the same oracle records")
    ap.add_argument("--no-pcie-pass", action="store_true")
    ap.add_argument("--no-bam-pass", action="store_true")
    ap.add_argument("--no-neighbors", action="store_true")
    ap.add_argument("--broadcast-index", dest="broadcast_index", action="store_true", default=True, help="with several GPUs: rank 0 builds the table, the others receive it over RCCL (the default)")
    ap.add_argument("--no-broadcast-index", dest="broadcast_index", action="store_false", help="with several GPUs: every rank builds its own table (N builds side by side)")
    ap.add_argument("--no-cli-pass", action="store_true", help="skip the isaac-align end-to-end leg (config.cli_end_to_end)")
    ap.add_argument("--cli-pairs", type=int, default=10_000_000, help="pairs the isaac-align leg aligns (the run's own batches, written as FASTQ)")
    ap.add_argument("--contexts", type=int, default=4, help="contexts (streams) per GPU that take the steps' selections in turn; they share the contigs and the table")
    ap.add_argument("--stream-lookups", action="store_true", help="select a step as soon as it is looked up once every contig has a match (isaac-align's closeHits) instead of all lookups first; measured: no gain with the reads resident -- a lookup that shares the GPU with three selections takes 5.4 ms instead of 1.15 and sits on every step's critical path (profiles/exp_r6_stream.log)")
    ap.add_argument("--no-single-stream-pass", action="store_true", help="skip the extra pass of the same steps on one context (per-kernel times without sharing)")
    ap.add_argument("--launch-check", action="store_true", help="GPU-less check of the launcher and the collectives (gloo): no alignment")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """N > 1 without a launcher: the ranks are started as a child process tree of this one (never an exec after GPU init)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


_RESULT_FD = None


def claim_stdout():
    """The result is the only thing this process writes to its standard output: libraries that print there (RCCL writes a version banner
    when a process group comes up) are sent to standard error, and emit() writes the one JSON line to the real descriptor."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(result):
    data = (json.dumps(result) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def launch_check(args, rank, world):
    """the run's three collectives on fake data over gloo (CPU): what tests/test_bench_launch.py drives"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from isaac_aligner_amd import abi, shard
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    hits = np.zeros(8, np.uint8); hits[rank % 8] = 1
    merged = shard.reduce_contig_hits(hits, dist)
    tls = abi.Tls(); tls.min, tls.max, tls.median = 100 + rank, 500 + rank, 300 + rank
    shard.broadcast_tls(tls, dist)
    begin, end = shard.shard_bounds(1001, rank, world)
    records = torch.full((end - begin, 4), rank, dtype=torch.uint8)
    got = shard.gather_records(records, dist, rank, world)
    counts = [shard.shard_bounds(1001, r, world)[1] - shard.shard_bounds(1001, r, world)[0] for r in range(world)]
    gatherer = shard.StepGather(dist, rank, world, record_counts=counts)       # what the timed loop does per step
    for step in range(2):
        pool = torch.full((4 * (end - begin),), 16 * rank + step, dtype=torch.int32)
        gatherer.add(records + step, pool, torch.tensor([3 * (end - begin) + rank], dtype=torch.int64))
    steps = gatherer.finish()
    dist.barrier()
    if rank == 0:
        ok = int(merged.sum()) == min(world, 8) and tls.min == 100 and sum(len(g) for g in got) == 1001 and all(int(g[0, 0]) == r for r, g in enumerate(got))
        ok = ok and len(steps) == 2 and all(sum(len(r_) for r_ in recs) == 1001 and all(int(c_[0]) == 16 * r + st and len(c_) == 3 * counts[r] + r for r, c_ in enumerate(cigs))
                                            for st, (recs, cigs) in enumerate(steps))
        emit({"launch_check": bool(ok), "n_gpus": world, "gpus_requested": args.gpus})
    dist.destroy_process_group()


def cli_end_to_end(args, al, genome, batches, L, emit_note):
    """The drop-in itself, end to end: bin/isaac-align (C++ on the C ABI) on the run's own reference and reads as files -- the synthetic genome as FASTA, the
    resident table as the 64 mask files + sorted-reference.xml of isaac-sort-reference (isaac_gpu_save_sorted_reference), the batches as two FASTQ files --
    with the reference's defaults (duplicates marked, gaps realigned, --bam-gzip-level 1).  Files in /dev/shm when there is one (47 GB of mask files).
    Returns the dict for config.cli_end_to_end: reads per second over the program's wall time and its own stage timers."""
    import json
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import torch
    from isaac_aligner_amd import build, sorted_reference as sr, synth
    t_all = time.perf_counter()
    work = tempfile.mkdtemp(prefix="isaac_bench_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        ref_dir, calls = os.path.join(work, "ref"), os.path.join(work, "calls")
        os.makedirs(ref_dir); os.makedirs(calls)
        fasta = os.path.join(ref_dir, "genome.fa")
        contigs, position = [], 0
        t0 = time.perf_counter()
        for i, (offset, size, bases, acgt) in enumerate(synth.write_fasta(fasta, genome.contigs)):
            m = sr.Contig()
            m.genomic_position, m.index, m.karyotype_index, m.name, m.file = position, i, i, b"chr%d" % (i + 1), fasta.encode()
            m.offset, m.size, m.total_bases, m.acgt_bases = offset, size, bases, acgt
            position += bases
            contigs.append(m)
        t_fasta = time.perf_counter() - t0
        t0 = time.perf_counter()
        al.save_sorted_reference(ref_dir, "genome.fa", contigs)
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        n_pairs, files = 0, [open(os.path.join(calls, "lane1_read%d.fastq" % (r + 1)), "wb") for r in range(2)]
        for batch in batches:
            if n_pairs >= args.cli_pairs:
                break
            bcl = batch[:args.cli_pairs - n_pairs].cpu().numpy()
            synth.write_fastq(files, bcl, L, name_prefix=b"M1:7:FCBENCH:1:1101:", first_index=n_pairs)
            n_pairs += len(bcl)
        for f in files:
            f.close()
        t_fastq = time.perf_counter() - t0
        tool = build.build_host()
        cmd = [tool, "-r", os.path.join(ref_dir, "sorted-reference.xml"), "-b", calls, "--base-calls-format", "fastq", "-o", os.path.join(work, "Aligned"), "--use-bases-mask", "y*,y*",
               "--clusters-at-a-time", "2000000"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode:
            return {"error": (r.stderr or r.stdout)[-600:]}
        timing = json.loads([l for l in r.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
        bam = os.path.join(work, "Aligned", "Projects", "default", "default", "sorted.bam")
        out = {"reads_per_s": round(timing["reads"] / wall, 1), "reads_per_s_without_reference_load": round(timing["reads"] / max(1e-9, timing["total_s"] - timing["reference_s"]), 1),
               "pairs": n_pairs, "wall_s": round(wall, 2), "stages_s": {k: (round(v, 3) if not isinstance(v, dict) else {kk: round(vv, 3) for kk, vv in v.items()}) for k, v in timing.items() if k.endswith("_s")}, "records": timing["records"], "workers": timing["workers"],
               "tiles_kept_on_device": timing.get("tiles_kept_on_device"),
               "sorted_bam_bytes": os.path.getsize(bam), "bai_bytes": os.path.getsize(bam + ".bai"),
               "command": "isaac-align -r sorted-reference.xml -b <2 FASTQ files> --base-calls-format fastq --use-bases-mask y*,y* --clusters-at-a-time 2000000 (defaults: --mark-duplicates 1, --realign-gaps sample, --bam-gzip-level 1)",
               "preparation_s": {"fasta": round(t_fasta, 1), "save_sorted_reference": round(t_save, 1), "fastq": round(t_fastq, 1)},
               "note": "the program's own wall time from start to finished sorted.bam + .bai, files in %s; reference_s is reading the FASTA and the 64 mask files (%.0f GB) back in" % (os.path.dirname(work), al_table_gb(al))}
        out["leg_s"] = round(time.perf_counter() - t_all, 1)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def al_table_gb(al):
    return 16.0 * al.index_tensors().shape[0] / 1e9


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    claim_stdout()
    if args.launch_check:
        return launch_check(args, rank, world)

    # The contexts' streams, torch's own and (with several GPUs) RCCL's want a hardware queue each; the runtime's default is four, and streams that
    # share a queue run one after the other (four contexts on four queues: 26.6 ms per step instead of 24.4).  Read when the runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    from isaac_aligner_amd import abi, gpu, options, shard, synth
    dist = None
    if world > 1 or os.environ.get("ISAAC_BENCH_FORCE_DIST"):      # the variable: run the collectives with one rank too (1-GPU check of the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = args.read_length
    params = options.default_params(L, L)
    read_kw = {}
    if args.indel_read_fraction is not None:
        read_kw["indel_read_fraction"] = args.indel_read_fraction
    if args.indel_max is not None:
        read_kw["indel_max"] = args.indel_max

    # ---- setup (untimed): reference, index, reads resident in HBM ---------------------------------------------------
    t0 = time.time()
    genome = synth.make_human_like_genome(args.genome_bases, seed=3, device=dev)
    torch.cuda.empty_cache()
    al = gpu.Aligner(params, local_rank, genome, deferred_completion=True)   # back-to-back select calls overlap; al.synchronize() completes them
    t_genome = time.time() - t0
    if dist is not None and args.broadcast_index:
        # one build, N - 1 transfers of the table (47 GB at this size) over xGMI
        if rank == 0:
            n_index = al.build_index(repeat_threshold=1000, annotate_neighbors=not args.no_neighbors)
        shared_table = shard.broadcast_table(al.index_tensors() if rank == 0 else None, dist, rank, dev)
        if rank != 0:
            al.set_index_tensors(shared_table)
            n_index = int(shared_table.shape[0])
    else:
        n_index = al.build_index(repeat_threshold=1000, annotate_neighbors=not args.no_neighbors)
    t_index = time.time() - t0 - t_genome
    # More contexts on the same GPU, each on a stream of its own, sharing the contigs and the resident table (isaac_gpu_set_index_dev): the
    # selections of consecutive steps are handed to them in turn, so that kernels of different steps share the GPU -- the tail of a launch,
    # the few long workgroups of the repeat-family sums and the kernels that wait on memory overlap another step's arithmetic.
    n_contexts = max(1, args.contexts)
    als, streams = [al], [torch.cuda.current_stream(dev)]
    table = al.index_tensors() if n_contexts > 1 else None
    for _ in range(n_contexts - 1):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            extra = gpu.Aligner(params, local_rank, genome, deferred_completion=True)
            extra.set_index_tensors(table)
        als.append(extra)
        streams.append(st)
    # ... and one for the lookups, so that the lookup of a later step does not queue behind a selection
    finder_stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(finder_stream):
        finder = gpu.Aligner(params, local_rank, genome)
        finder.set_index_tensors(al.index_tensors())
    finders, finder_streams = [finder], [finder_stream]
    for _ in range(1, max(1, args.finders)):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            extra = gpu.Aligner(params, local_rank, genome)
            extra.set_index_tensors(al.index_tensors())
        finders.append(extra)
        finder_streams.append(st)
    n_batches = args.warmup + args.steps
    per_rank = args.pairs_per_step
    batches = []
    for b in range(n_batches):
        if args.scaling == "strong" and b >= args.warmup:   # one read set for the whole job, cut N ways (every rank draws it, keeps its shard)
            bcl = synth.make_read_pairs(genome, args.pairs_per_step, L, seed=1000 + b, device=dev, avoid_gaps=True, **read_kw)[0]
            begin, end = shard.shard_bounds(args.pairs_per_step, rank, world)
            bcl = bcl[begin:end].contiguous()
            per_rank = end - begin
        else:
            bcl = synth.make_read_pairs(genome, args.pairs_per_step, L, seed=1000 * (rank + 1) + b, device=dev, avoid_gaps=True, **read_kw)[0]
        batches.append(bcl)
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    def buffers(n_pairs):
        n_rec = n_pairs * 2
        return (torch.empty((n_rec, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=dev), torch.empty(n_rec * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=dev),
                torch.empty(n_rec * 4, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    out = [buffers(batches[args.warmup + s].shape[0]) for s in range(args.steps)]     # records, 40-word CIGAR slots, packed CIGARs of every step + their length
    # ... and the match lists of every step (the hand-over between the two phases): nothing is allocated inside the timed steps
    match_bufs = [(torch.empty((al.match_capacity(batches[args.warmup + s].shape[0]), 2), dtype=torch.int64, device=dev),
                   torch.empty(batches[args.warmup + s].shape[0] + 1, dtype=torch.int64, device=dev)) for s in range(args.steps)]
    tile_of = lambda s: 1 + rank * args.steps + s          # every (rank, step) batch is a tile of its own, as FASTQ tiles are (SeedId.hh: 12 bits)

    def reduce_hits(h):
        return shard.reduce_contig_hits(h, dist, dev)

    # ---- warm-up: also learns the template length statistics from the first batch (as tile 1 of the reference does) ---
    tls = None
    for b in range(max(1, args.warmup)):
        m, o, hits = al.find_matches(batches[b])
        warm_loaded = reduce_hits(hits)
        al.set_loaded_contigs(warm_loaded)
        if tls is None:
            tls = al.determine_tls(batches[b], m, o)
        if b < args.warmup:
            w = buffers(batches[b].shape[0])
            for ctx in als:                               # every context grows its chunk buffers here, not in the timed steps
                if ctx is not al:
                    ctx.set_loaded_contigs(warm_loaded)
                ctx.select(batches[b], m, o, tls, out=w[:2])
                ctx.synchronize()
                ctx.compact_cigars(w[0], w[1], w[2])
            del w
    shard.broadcast_tls(tls, dist, dev)   # rank 0's statistics are the run's statistics
    if dist is not None:
        # the first gather of a process group sets up its point-to-point channels (most of a second): not part of the steps
        warm = shard.StepGather(dist, rank, world)
        warm.add(torch.zeros((64, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=dev), torch.zeros(256, dtype=torch.int32, device=dev), torch.full((1,), 200, dtype=torch.int64, device=dev))
        warm.finish()
        del warm
        if rank == 0:
            # rank 0 receives world x (records + packed CIGARs) per step: the blocks are taken from the driver once, here, and handed back to
            # torch's caching allocator, so that the timed steps find them there (a fresh hipMalloc of 100s of MB costs milliseconds)
            pool = [torch.empty_like(out[s][0]) for s in range(args.steps) for _ in range(world)]
            pool += [torch.empty_like(out[s][2]) for s in range(args.steps) for _ in range(world)]
            del pool
    for f, st in zip(finders, finder_streams):       # (the lookup contexts grow their buffers here, not in the timed steps)
        with torch.cuda.stream(st):
            f.find_matches(batches[0], tile=tile_of(0), out=match_bufs[0])
        f.synchronize(); f.reset_timers()
    for ctx in als:
        ctx.synchronize()
        ctx.reset_timers()

    # ---- timed region: exactly K steps ---------------------------------------------------------------------------------
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    # FindMatchesTransition comes before SelectMatchesTransition because the selection wants to know which contigs have matches anywhere in the run
    # (MatchSelector loads only those: the rest-of-genome correction, hence MAPQ, depends on the set): all lookups, then all selections.
    # --stream-lookups: once every contig has a match -- a whole-genome run gets there with its first tile -- no later lookup can change that set, and from
    # there on a step is selected as soon as it is looked up, the later lookups (a context and stream of their own) beside the selections, as isaac-align itself
    # does (closeHits).  Same records; with the reads resident it is no faster (see the option's help), so the two phases stay apart by default.  With several GPUs the contig flags are
    # all-reduced after every lookup until the set is complete -- every rank sees the same reduced set, so every rank takes the same turn.
    # Every step's CIGARs are packed behind its selection without a host wait (isaac_gpu_compact_cigars_async: the packed length stays on the
    # device); with several GPUs the step's records and CIGAR pool then leave for rank 0 behind the later steps (shard.StepGather: nothing
    # in the loop waits for the GPU or for another rank -- every rank's record count is known from the static split)
    if dist is not None:
        if args.scaling == "strong":
            rec_counts = [2 * (shard.shard_bounds(args.pairs_per_step, r, world)[1] - shard.shard_bounds(args.pairs_per_step, r, world)[0]) for r in range(world)]
        else:
            rec_counts = [2 * args.pairs_per_step] * world
    gatherer = shard.StepGather(dist, rank, world, record_counts=rec_counts) if dist is not None else None
    # the gathers are issued from a stream of their own that waits for the step's context: issued from the stream a context computes on, that
    # context's later steps would queue behind the other contexts' steps and the contexts would take turns instead of overlapping
    gather_stream = torch.cuda.Stream(device=dev) if gatherer is not None else None
    found, looked_up = [], []
    all_hits = np.zeros(al.n_contigs, np.uint8)
    loaded, streaming, streaming_from, next_select = None, False, None, 0

    def select_step(s):                               # SelectMatchesTransition for step s, on the context whose turn it is
        m, o = found[s]
        ctx = als[s % n_contexts]
        ctx_stream = streams[s % n_contexts]
        ctx_stream.wait_event(looked_up[s])
        with torch.cuda.stream(ctx_stream):
            ctx.select(batches[args.warmup + s], m, o, tls, tile=tile_of(s), out=out[s][:2])
            ctx.compact_cigars_async(out[s][0], out[s][1], out[s][2], out[s][3])
        if gatherer is not None:
            step_done = ctx_stream.record_event()
            with torch.cuda.stream(gather_stream):
                gather_stream.wait_event(step_done)
                gatherer.add(out[s][0], out[s][2], out[s][3])
    def look_up(s):                                   # FindMatchesTransition for step s, on the lookup context whose turn it is
        torch.cuda.set_device(dev)
        f, st = finders[s % len(finders)], finder_streams[s % len(finders)]
        with torch.cuda.stream(st):
            m, o, hits = f.find_matches(batches[args.warmup + s], tile=tile_of(s), out=match_bufs[s])
            return m, o, hits, st.record_event()
    ahead = None
    if len(finders) > 1 and not args.stream_lookups:
        # the two phases apart (the default): the lookups of the steps side by side, a host thread per lookup context (the calls release the interpreter)
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(len(finders))
        ahead = [pool.submit(look_up, s) for s in range(args.steps)]
    for s in range(args.steps):
        m, o, hits, event = ahead[s].result() if ahead is not None else look_up(s)
        looked_up.append(event)
        found.append((m, o))
        all_hits |= hits
        if not streaming:
            if not args.stream_lookups:
                continue                              # (one exchange of the flags behind the last lookup, below)
            loaded = reduce_hits(all_hits)
            if loaded.all():
                streaming, streaming_from = True, s
                for ctx in als:
                    ctx.set_loaded_contigs(loaded)    # MatchSelector loads only contigs that received matches: all of them from here on
        while streaming and next_select <= s:
            select_step(next_select); next_select += 1
    t_lookups_done = time.perf_counter() - t_start    # (host time: every lookup has returned its contig flags by now)
    if not streaming:
        loaded = reduce_hits(all_hits)
        for ctx in als:
            ctx.set_loaded_contigs(loaded)
    while next_select < args.steps:
        select_step(next_select); next_select += 1
    if ahead is not None:
        pool.shutdown()
    for f in finders:
        f.synchronize()
    for ctx in als:
        ctx.synchronize()
    gathered = gatherer.finish() if gatherer is not None else None
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    timed_streaming_from = streaming_from
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n_pairs_all = torch.tensor([sum(b.shape[0] for b in batches[args.warmup:])], dtype=torch.int64, device=dev)
        dist.all_reduce(n_pairs_all)
        pairs_total = int(n_pairs_all.item())
    else:
        pairs_total = sum(b.shape[0] for b in batches[args.warmup:])

    packed = [out[s][2][:int(out[s][3].item())] for s in range(args.steps)]
    assert all(int(out[s][3].item()) <= out[s][2].numel() for s in range(args.steps)), "a CIGAR pool of 4 words per record was too small"
    # the first timed step's result, kept for the parity check below (the PCIe-inclusive pass writes the same buffers again)
    checked_records, checked_cigars = out[0][0].clone(), packed[0].clone()
    gather_identical = None
    if gathered is not None and rank == 0:          # rank 0's own step as it came back through the gather
        gather_identical = bool((gathered[0][0][0] == checked_records).all()) and bool((gathered[0][1][0] == checked_cigars).all())
    del gathered
    # clusters whose MAPQ arithmetic sat within 1e-11 of an integer (isaac_fragment::reserved bit 3), of every step: re-derived by the oracle below
    flagged = []
    for s in range(args.steps):
        pairs_s = torch.nonzero((out[s][0].view(torch.int32)[:, 15] & 8) != 0).flatten() // 2
        pairs_s = torch.unique(pairs_s)
        if pairs_s.numel():
            flagged.append((s, batches[args.warmup + s][pairs_s].cpu().numpy(),
                            out[s][0].view(-1, 2, abi.FRAGMENT_DTYPE.itemsize)[pairs_s].reshape(-1, abi.FRAGMENT_DTYPE.itemsize).cpu().numpy()))
    flagged_cigars = {s: packed[s].cpu().numpy().view(np.uint32) for s, _, _ in flagged}
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)      # everything resident at the end of the run: index, reads, records, chunk scratch
    counters = al.counters()
    for ctx in als[1:] + finders:                    # (the lookups' probes and matches are counted by the context that ran them)
        for key, value in ctx.counters().items():
            counters[key] += value
    timer_names = ("find_matches", "compact_matches", "build_fragments", "build_fragments_general", "align_candidates", "finish_candidates", "finish_candidates_general", "indel_fragments", "gapped_fragments",
                   "gapped_fragments_rescan", "finish_fragments", "finish_fragments_general",
                   "plan_rescue", "rescue_windows", "rescue_align", "rescue_gapped_plan", "gapped_rescue", "gapped_rescue_rescan", "sums_wave", "sums_large", "sums_xl", "sums_huge", "select",
                   "select_heavy", "select_residual")
    def read_timers(contexts):
        t = {}
        for k in timer_names:                         # average duration and launches over all contexts
            per = [ctx.kernel_time_ms(k) for ctx in contexts]
            launches = sum(n for _, n in per)
            t[k] = (sum(ms * n for ms, n in per) / launches if launches else 0.0, launches)
        return t
    timers = read_timers(als + finders)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    reads_per_s = 2.0 * pairs_total / elapsed
    pairs_rank = sum(b.shape[0] for b in batches[args.warmup:])

    # ---- the same K selections on ONE context (not `value`): with several contexts the kernels of different steps share the GPU in the
    # timed region and a kernel's event time includes what it shares; here every kernel has the GPU to itself.  Also shows that the
    # records do not depend on how the steps were interleaved.
    single = None
    if n_contexts > 1 and dist is None and not args.no_single_stream_pass:
        al.reset_timers()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(args.steps):
            al.select(batches[args.warmup + s], found[s][0], found[s][1], tls, tile=tile_of(s), out=out[s][:2])
            al.compact_cigars_async(out[s][0], out[s][1], out[s][2], out[s][3])
        al.synchronize()
        torch.cuda.synchronize()
        single_elapsed = time.perf_counter() - t1
        single = {"select_ms_per_step": round(single_elapsed / args.steps * 1e3, 3), "timers": read_timers([al]),
                  "records_identical_to_timed_region": bool((out[0][0] == checked_records).all()) and bool((packed[0] == checked_cigars).all())}

    # ---- the same K steps with the reads arriving over PCIe and the results leaving over it (reported next to `value`, never as it)
    pcie = None
    if not args.no_pcie_pass and dist is None:
        host_in = [torch.empty(b.shape, dtype=torch.uint8).pin_memory() for b in batches[args.warmup:]]
        for h, b in zip(host_in, batches[args.warmup:]):
            h.copy_(b)
        host_rec = [torch.empty(o_[0].shape, dtype=torch.uint8).pin_memory() for o_ in out]
        host_cig = [torch.empty(p_.shape, dtype=torch.int32).pin_memory() for p_ in packed]
        copy_stream = torch.cuda.Stream(dev)
        # (the lookups have the context and stream of their own that the timed region gave them)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        dev_in = []
        for s in range(args.steps):
            with torch.cuda.stream(copy_stream):          # uploads run ahead of the lookups on their own stream
                d = host_in[s].to(dev, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(copy_stream)
            dev_in.append((d, ev))
        host_n = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(args.steps)]
        left = [None] * args.steps

        def download_cigars(k):
            left[k].synchronize()                         # step k is complete (the steps after it are queued already: the GPU stays busy)
            n_k = int(host_n[k].item())
            with torch.cuda.stream(copy_stream):
                host_cig[k][:n_k].copy_(out[k][2][:n_k], non_blocking=True)

        def select_step(k):
            m, o = found_p[k]
            ctx = als[k % n_contexts]                     # the selections dealt to the contexts in turn, as in the timed region
            ctx_stream = streams[k % n_contexts] if ctx is not al else torch.cuda.current_stream(dev)
            ctx_stream.wait_event(looked_up[k])
            ctx.select(dev_in[k][0], m, o, tls, tile=tile_of(k), out=out[k][:2])
            ctx.compact_cigars_async(out[k][0], out[k][1], out[k][2], out[k][3])
            done = ctx_stream.record_event()
            if k >= n_contexts:
                download_cigars(k - n_contexts)           # the step this context did before: complete by now, or soon; the other contexts keep the GPU busy
            with torch.cuda.stream(copy_stream):          # the step's records and the length of its CIGAR pool leave as soon as they are final
                copy_stream.wait_event(done)
                host_rec[k].copy_(out[k][0], non_blocking=True)
                host_n[k].copy_(out[k][3], non_blocking=True)
                left[k] = torch.cuda.Event(); left[k].record(copy_stream)
        # FindMatchesTransition comes before SelectMatchesTransition because the selection wants to know which contigs have matches anywhere in the run.
        # Once every contig has one, no later lookup can change that set: from there on a step is selected as soon as it is looked up, and the uploads
        # and lookups of the later steps run beside the selections of the earlier ones (a whole-genome run gets there with its first tile)
        found_p, looked_up, hits_p, streaming, next_select, streaming_from = [], [], np.zeros(al.n_contigs, np.uint8), False, 0, None
        for s in range(args.steps):
            d, ev = dev_in[s]
            with torch.cuda.stream(finder_stream):
                finder_stream.wait_event(ev)
                m, o, hits = finder.find_matches(d, tile=tile_of(s), out=match_bufs[s])
                looked_up.append(finder_stream.record_event())
            found_p.append((m, o))
            hits_p |= hits
            if not streaming and hits_p.all():
                streaming, streaming_from = True, s
                for ctx in als:
                    ctx.set_loaded_contigs(hits_p)
            while streaming and next_select <= s:
                select_step(next_select); next_select += 1
        if not streaming:
            for ctx in als:
                ctx.set_loaded_contigs(hits_p)
        while next_select < args.steps:
            select_step(next_select); next_select += 1
        for k in range(max(0, args.steps - n_contexts), args.steps):
            download_cigars(k)
        for ctx in als:
            ctx.synchronize()
        torch.cuda.synchronize()
        t_pcie = time.perf_counter() - tp
        same = bool((out[0][0] == checked_records).all()) and bool((out[0][2][:checked_cigars.numel()] == checked_cigars).all())
        pcie = {"reads_per_s": round(2.0 * pairs_rank / t_pcie, 1), "records_identical_to_resident_pass": same, "ms_per_step": round(1e3 * t_pcie / args.steps, 3),
                "bytes_in_per_pair": 2 * L, "bytes_out_per_pair": round((sum(r.numel() for r in host_rec) + 4 * sum(int(p.numel()) for p in packed)) / pairs_rank, 1),
                "contexts": n_contexts,
                "selections_began_after_lookup_of_step": streaming_from,
                "note": "BCL bytes uploaded from pinned host memory ahead of the lookups (a context and stream of their own); once every contig has a match -- after the first step's lookup here -- no later lookup can change the set of loaded contigs, and a step is selected as soon as it is looked up, the later steps' uploads and lookups running beside it; the selections dealt to the contexts in turn as in the timed region; a step's records + packed CIGARs are downloaded on a copy stream while the steps queued behind it compute"}

    # ---- the output side: all steps' records as one position-sorted BAM record stream, resident in HBM; reported beside `value`
    bam_info = None
    if not args.no_bam_pass and dist is None:
        tiles = [(batches[args.warmup + s], out[s][0], out[s][2], "SYNTH:1:%d:" % tile_of(s)) for s in range(args.steps)]
        bam_buf = torch.empty(sum(o_[0].shape[0] for o_ in out) * (100 + L + (L + 1) // 2 + 8), dtype=torch.uint8, device=dev)
        al.bam_records(tiles, out=bam_buf)               # warm-up: scratch allocation
        al.reset_timers()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        stream_bytes, n_bam, unaligned_at = al.bam_records(tiles, out=bam_buf)
        torch.cuda.synchronize()
        t_bam = time.perf_counter() - tb
        bam_info = {"records": int(n_bam), "bytes": int(stream_bytes.numel()), "ms": round(1e3 * t_bam, 3), "records_per_s": round(n_bam / t_bam, 1),
                    "GB_per_s_written": round(stream_bytes.numel() / t_bam / 1e9, 2), "order_ms": round(al.kernel_time_ms("bam_order")[0], 3),
                    "encode_ms": round(al.kernel_time_ms("bam_encode")[0], 3), "unaligned_bin_offset": int(unaligned_at),
                    "note": "isaac_gpu_bam_records over the records of all %d steps (two radix passes + one encode launch)" % args.steps}
        # --bam-gzip-level 0 entirely on the device: BGZF framing with stored blocks, CRC-32 per block computed by the GPU
        framed = al.bgzf_store(stream_bytes, eof_block=True)
        al.reset_timers()
        framed = al.bgzf_store(stream_bytes, eof_block=True, out=framed)
        torch.cuda.synchronize()
        t_store = al.kernel_time_ms("bgzf_store")[0]
        bam_info.update({"bgzf_store_ms": round(t_store, 3), "bgzf_store_GB_per_s": round(stream_bytes.numel() / max(1e-9, t_store) / 1e6, 1), "bgzf_store_bytes": int(framed.numel())})
        del framed
        # --bam-gzip-level 1 (the reference's default) on the device: isaac_gpu_bgzf_deflate over the whole record stream
        deflated = al.bgzf_deflate(stream_bytes, eof_block=True)           # warm-up: staging allocation
        deflate_out = torch.empty(int(deflated.numel()) + (1 << 20), dtype=torch.uint8, device=dev)
        del deflated
        torch.cuda.synchronize()
        td = time.perf_counter()
        deflated = al.bgzf_deflate(stream_bytes, eof_block=True, out=deflate_out)
        torch.cuda.synchronize()
        t_deflate = time.perf_counter() - td
        import zlib
        check_bytes = min(int(stream_bytes.numel()), 64 << 20) // 65494 * 65494          # whole blocks: the first of the deflated stream inflate to the first of the records
        dz = zlib.decompressobj(31)
        head = deflated[:check_bytes // 2 + (4 << 20)].cpu().numpy().tobytes()
        inflated = bytearray()
        while len(inflated) < check_bytes and head:
            inflated += dz.decompress(head)
            head = dz.unused_data
            if dz.eof:
                dz = zlib.decompressobj(31)
        bam_info.update({"bgzf_deflate_ms": round(1e3 * t_deflate, 3), "bgzf_deflate_GB_per_s": round(stream_bytes.numel() / t_deflate / 1e9, 2), "bgzf_deflate_bytes": int(deflated.numel()),
                         "bgzf_deflate_ratio": round(deflated.numel() / stream_bytes.numel(), 4), "bgzf_deflate_checked_bytes": check_bytes,
                         "bgzf_deflate_inflates_to_records": bytes(inflated[:check_bytes]) == stream_bytes[:check_bytes].cpu().numpy().tobytes()})
        del deflated, deflate_out, inflated
        # the host side of the file writer: BGZF deflate (zlib level 1, as --bam-gzip-level defaults) of a bounded sample on all host threads
        from isaac_aligner_amd import bam as bam_host
        sample_bytes = min(int(stream_bytes.numel()), 512 << 20)
        host_sample = stream_bytes[:sample_bytes].cpu().numpy()
        tz = time.perf_counter()
        z = bam_host.bgzf_compress(host_sample, level=1, n_threads=os.cpu_count() or 1)
        t_z = time.perf_counter() - tz
        bam_info.update({"bgzf_sample_bytes": sample_bytes, "bgzf_threads": os.cpu_count() or 1, "bgzf_GB_per_s": round(sample_bytes / t_z / 1e9, 2),
                         "bgzf_ratio": round(len(z) / sample_bytes, 3)})
        del bam_buf, stream_bytes, host_sample, z

    # ---- isaac-align itself, end to end (config.cli_end_to_end)
    cli_info = None
    if not args.no_cli_pass and dist is None and rank == 0:
        try:
            # the program gets the device as a run of its own would find it, beside this process's table and reads: the extra contexts (their timers
            # and records have been read) and torch's cached blocks go first
            for extra in als[1:] + finders:
                extra.close()
            del als[1:]
            torch.cuda.empty_cache()
            cli_info = cli_end_to_end(args, al, genome, batches[args.warmup:], L, None)
        except Exception as e:      # the leg must not cost the run its line
            cli_info = {"error": repr(e)[:400]}

    # ---- roofline of the dominant kernel: algorithmic bytes (stated per kernel in DESIGN.md) / event-timed duration
    c = counters
    seeded_scans = max(0, c["ungapped_scans"] - c["rescue_candidates"])
    jobs = c["rescue_calls"]
    per_kernel_bytes = {
        # BCL in + 16 B per probe step actually taken (prefix directory entry or table entry) + match records out
        "find_matches": 2 * L * pairs_rank + c["probe_steps"] * 16 + c["matches"] * 16,
        # match records in + BCL + candidate records out
        "build_fragments": c["matches"] * 16 + 2 * L * pairs_rank + c["candidates"] * 64,
        # per seeded candidate: the read + an L-base reference window in, the candidate record + 3 cigar words out
        "align_candidates": seeded_scans * (2 * L + 64 + 12),
        "finish_candidates": c["candidates"] * 128,
        "indel_fragments": c["simple_indels"] * (2 * L + 256 + 128),
        # per banded Smith-Waterman problem: job record in, read + (L + 15)-base reference window in, result record out
        "gapped_fragments": c["bsw_jobs"] * (80 + 2 * L + 15 + 232),
        "gapped_rescue": c["rescue_bsw"] * (80 + 2 * L + 15 + 232),
        "finish_fragments": c["candidates"] * 128 + c["bsw_jobs"] * 232,
        "plan_rescue": c["candidates"] * 64 + jobs * 72,
        # the mate's bases + the window bases (2 bits + 1 bit each) in, candidate start positions out
        "rescue_windows": jobs * (72 + L) + c["rescue_window_bases"] * 3 // 8 + c["rescue_candidates"] * 8,
        # per candidate start: the mate (L BCL bytes) + L reference bytes in, one candidate record + 3 cigar words out
        "rescue_align": c["rescue_candidates"] * (2 * L + 64 + 12),
        "rescue_gapped_plan": c["rescue_candidates"] * 64 + jobs * 72,
        # per rescued shadow: its candidate record in (twice: its own list and the pair list), 32 B of sums per cluster out
        "sums_wave": c["rescue_candidates"] * 2 * 64 + jobs * 96 + pairs_rank * 32,
        # seeded + rescued candidate records in, 2 FragmentHeader records + cigars out
        "select": c["candidates"] * 64 + c["rescue_candidates"] * (64 + 12) + 2 * pairs_rank * (64 + 4 * 3),
        "select_heavy": c["heavy_clusters"] * 64 * 1000,
    }
    total_ms = {k: v[0] * v[1] for k, v in timers.items()}
    # the dominant kernel: the one that takes longest with the GPU to itself when that pass was run -- in the timed region the kernels of several
    # steps share the GPU and their event times depend on who they happened to share it with (two runs named two different kernels)
    if single is not None:
        dominant = max(per_kernel_bytes, key=lambda k: single["timers"].get(k, (0.0, 0))[0] * single["timers"].get(k, (0.0, 0))[1])
    else:
        dominant = max(per_kernel_bytes, key=lambda k: total_ms.get(k, 0.0))
    launches = max(1, timers[dominant][1])
    avg_s = total_ms[dominant] / launches / 1e3
    achieved = per_kernel_bytes[dominant] / launches / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process.  The figure is taken from a
    # committed rocprofv3 --pmc summary (scripts/pmc_traffic.sh + pmc_summary.py) only when that summary was made on this
    # workload at this many pairs per launch; otherwise null.
    traffic, traffic_source = None, "no profiles/*_pmc_summary.json for this workload (genome_bases, read_length, pairs_per_launch)"
    prof_dir = os.path.join(ROOT, "profiles")
    pmc, pmc_name = None, None            # the newest counter summary of this workload (what north_star's rooflines below are made of)
    for name in sorted((n for n in os.listdir(prof_dir) if n.endswith("_pmc_summary.json")), reverse=True) if os.path.isdir(prof_dir) else []:
        summary = json.load(open(os.path.join(prof_dir, name)))
        w = summary.get("workload", {})
        if pmc is None and w.get("genome_bases") == args.genome_bases and w.get("read_length") == L and w.get("pairs_per_launch") == args.pairs_per_step:
            pmc, pmc_name = summary, name
        # per (kernel, timer): a kernel that several stages launch has an entry per stage where the summary makes the split
        k = summary.get("k_" + dominant) or (summary.get("k_gapped_jobs:fragments") if dominant == "gapped_fragments" else None)
        if k and "hbm_bytes_per_launch" in k and w.get("genome_bases") == args.genome_bases and w.get("read_length") == L and w.get("pairs_per_launch") == args.pairs_per_step:
            traffic, traffic_source = int(k["hbm_bytes_per_launch"]), "profiles/%s (separate rocprofv3 --pmc passes over the same workload, per launch)" % name
            break
    # SURVEY.md §8d's bytes per pair with the run's own counters (probe steps as measured, not ceil(log2 n))
    bytes_pair = (2 * L + (c["probe_steps"] * 16 + c["matches"] * 16 + seeded_scans * (L + 15) + c["bsw_jobs"] * (L + 15) + c["rescue_window_bases"] +
                           c["rescue_candidates"] * L) / pairs_rank + 2 * (64 + 12))
    elapsed_rank = elapsed
    # the kernel behind a timer, where the names differ (the banded Smith-Waterman kernel serves the fragment stage and the mate rescue)
    kernel_of = {"gapped_fragments": "k_gapped_jobs", "gapped_rescue": "k_gapped_jobs", "sums_wave": "k_cluster_sums16"}
    single_stream = None
    if single is not None:      # the dominant kernel and every kernel's time with the GPU to itself
        st = single["timers"]
        st_launches = max(1, st[dominant][1])
        st_achieved = per_kernel_bytes[dominant] / st_launches / (st[dominant][0] / 1e3) / 1e9 if st[dominant][0] > 0 else 0.0
        single_stream = {"avg_launch_ms": round(st[dominant][0], 4), "achieved": round(st_achieved, 3), "frac": round(st_achieved / 8000.0, 6),
                         "select_ms_per_step": single["select_ms_per_step"], "records_identical_to_timed_region": single["records_identical_to_timed_region"],
                         "kernel_ms_per_step": {k: round(v[0] * v[1] / args.steps, 3) for k, v in st.items() if v[0] * v[1]},
                         "band_cell_updates_per_s": round((c["bsw_jobs"] + c["rescue_bsw"]) * L * 16 / max(1e-9, (st["gapped_fragments"][0] * st["gapped_fragments"][1] + st["gapped_rescue"][0] * st["gapped_rescue"][1]) / 1e3), 1)}
    roofline = {"bound": "hbm", "kernel": kernel_of.get(dominant, "k_" + dominant), "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": round(total_ms[dominant] / launches, 4), "launches": int(launches),
                "algorithmic_bytes_per_launch": int(per_kernel_bytes[dominant] / launches),
                "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in total_ms.items() if v},
                "bytes_per_pair": round(bytes_pair, 1),
                # the whole path of this GPU against the same peak (pairs/s x algorithmic bytes per pair)
                "path_achieved": round(pairs_rank * bytes_pair / elapsed_rank / 1e9, 2),
                "path_frac": round(pairs_rank * bytes_pair / elapsed_rank / 1e9 / 8000.0, 6),
                "heavy_clusters": int(c.get("heavy_clusters", 0)),
                # how many steps' kernels share the GPU in the timed region (a kernel's event time then includes what it shares), and the same
                # kernels with the GPU to themselves
                "concurrent_contexts": n_contexts, "single_stream": single_stream,
                # the banded Smith-Waterman kernels are VALU-bound: 16 band cells per row and problem (report cell updates/s)
                "band_cell_updates_per_s": round((c["bsw_jobs"] + c["rescue_bsw"]) * L * 16 / max(1e-9, (total_ms.get("gapped_fragments", 0.0) + total_ms.get("gapped_rescue", 0.0)) / 1e3), 1)}

    # ---- what north_star asks to be shown beside the number: achieved HBM GB/s of the seed lookup, LDS / occupancy of the banded SW, and -- the ceiling that
    # binds an int16 DP and this step -- vector issue.  Durations are this run's (HIP events; the kernels alone where the single-stream pass ran); the counters
    # are the committed summary's of the same workload (separate rocprofv3 --pmc passes: counters cannot be read from inside this process).
    SIMDS, CLOCK_HZ = 1024, 2.4e9                    # 256 CUs x 4 SIMDs; one wave instruction issues in 4 clocks (MI355X_MICROARCH.md)
    alone = single["timers"] if single is not None else timers
    def duration_s(timer):                            # average launch of a timer's kernel with the GPU to itself, seconds
        return alone.get(timer, (0.0, 0))[0] / 1e3
    find_launches = max(1, timers["find_matches"][1])
    find_s = timers["find_matches"][0] / 1e3         # (the lookups are not part of the single-stream pass: they run one after the other in the timed region too)
    find_algorithmic = per_kernel_bytes["find_matches"] / find_launches
    seed_lookup = {"kernel": "k_find_matches", "avg_launch_ms": round(find_s * 1e3, 4), "algorithmic_bytes_per_launch": int(find_algorithmic),
                   "algorithmic_GBps": round(find_algorithmic / find_s / 1e9, 1) if find_s else None, "frac": round(find_algorithmic / find_s / 1e9 / 8000.0, 5) if find_s else None,
                   "line_GBps": None, "traffic_ratio": None}
    sw = {"kernel": "k_gapped_jobs", "cells_per_s": (single_stream or roofline)["band_cell_updates_per_s"], "lanes_per_problem": 8, "lds_bytes_per_wg": None, "waves_per_simd": None,
          "lds_bank_conflict_frac": None, "valu_issue_frac": None}
    # LDS per workgroup (csrc/kernels.h: gappedGroupLdsBytes, eight problems per wavefront and workgroup) and what it lets a CU hold
    sw["lds_bytes_per_wg"] = 8 * ((((128 + (L + 7) // 8 * 80 + 15) // 16) | 1) * 16) if L <= 305 else None
    vgprs = None
    for name in sorted((n for n in os.listdir(prof_dir) if n.endswith("kernel_resources.txt")), reverse=True) if os.path.isdir(prof_dir) else []:
        for line in open(os.path.join(prof_dir, name)):
            f = line.split()
            if len(f) >= 3 and f[0].endswith("k_gapped_jobs" if L <= 177 else "k_gapped_jobs_long"):
                vgprs = int(f[2])
        if vgprs:
            sw["vgprs"], sw["resources_source"] = vgprs, "profiles/" + name
            break
    if sw["lds_bytes_per_wg"]:
        by_lds = (160 * 1024 // sw["lds_bytes_per_wg"]) / 4.0
        sw["waves_per_simd"] = min(by_lds, float(512 // vgprs)) if vgprs else by_lds
    valu_issue = {"step": None, "dominant": None}
    if pmc is not None:
        fm = pmc.get("k_find_matches", {})
        if fm.get("hbm_bytes_per_launch") and find_s:
            seed_lookup["line_GBps"] = round(fm["hbm_bytes_per_launch"] / find_s / 1e9, 1)
            seed_lookup["traffic_ratio"] = round(fm["hbm_bytes_per_launch"] / find_algorithmic, 3)
            seed_lookup["valu_lane_utilisation"] = round(fm.get("valu_lane_utilisation", 0.0), 3)
        gj = pmc.get("k_gapped_jobs:fragments") or pmc.get("k_gapped_jobs", {})
        if gj.get("lds_bank_conflict_frac") is not None:
            sw["lds_bank_conflict_frac"] = round(gj["lds_bank_conflict_frac"], 4)
        def issue_s(entry):                           # seconds the chip's SIMDs need to issue one launch's vector instructions
            return entry["SQ_INSTS_VALU"] / entry["launches"] * 4.0 / (SIMDS * CLOCK_HZ) if entry.get("SQ_INSTS_VALU") and entry.get("launches") else 0.0
        if gj.get("SQ_INSTS_VALU") and duration_s("gapped_fragments"):
            sw["valu_issue_frac"] = round(issue_s(gj) / duration_s("gapped_fragments"), 3)
        # the step: every kernel of one lookup + one selection, launches per step as the timed region made them
        step_kernels = {"find_matches": ["k_find_matches"], "compact_matches": ["k_compact_matches"], "build_fragments": ["k_build_fragments"], "build_fragments_general": ["k_build_fragments_general"],
                        "align_candidates": ["k_align_candidates"], "finish_candidates": ["k_finish_candidates", "k_cluster_kinds", "k_cluster_order"], "finish_candidates_general": ["k_finish_candidates_general"],
                        "indel_fragments": ["k_indel_fragments"], "gapped_fragments": ["k_gapped_jobs:fragments"], "gapped_fragments_rescan": ["k_gapped_rescan"], "finish_fragments": ["k_finish_fragments"],
                        "finish_fragments_general": ["k_finish_fragments_general"], "plan_rescue": ["k_plan_rescue"], "rescue_windows": ["k_rescue_windows"], "rescue_align": ["k_rescue_align"],
                        "rescue_gapped_plan": ["k_rescue_gapped_plan", "k_rescue_gapped_plan_long"], "gapped_rescue": ["k_gapped_jobs:rescue"], "gapped_rescue_rescan": ["k_gapped_rescan"],
                        "sums_wave": ["k_cluster_sums16", "k_cluster_sums"], "sums_large": ["k_cluster_sums_mid", "k_cluster_sums_large"], "sums_xl": ["k_cluster_sums_xl"], "sums_huge": ["k_cluster_sums_huge"],
                        "select": ["k_select"]}
        step_issue = sum(issue_s(pmc[k]) for names in step_kernels.values() for k in names if k in pmc)
        dom_names = step_kernels.get(dominant, [])
        dom_issue = sum(issue_s(pmc[k]) for k in dom_names if k in pmc)
        valu_issue = {"step": round(step_issue / (elapsed / args.steps), 3) if step_issue else None, "step_issue_floor_ms": round(step_issue * 1e3, 3),
                      "dominant": round(dom_issue / duration_s(dominant), 3) if dom_issue and duration_s(dominant) else None,
                      "dominant_issue_floor_ms": round(dom_issue * 1e3, 3), "source": "profiles/" + pmc_name,
                      "note": "sum over the kernels of SQ_INSTS_VALU per launch x 4 clocks / (1024 SIMDs x 2.4 GHz), over the measured time: the step's on the timed region, the dominant kernel's with the GPU to itself"}
    roofline["seed_lookup"] = seed_lookup
    roofline["sw"] = sw
    roofline["valu_issue_frac"] = valu_issue

    # ---- CPU baseline + parity: the oracle (a port of the reference path) on a bounded sample of the same workload, host cores ---
    cpu, parity = None, {"parity_checked_pairs": 0, "parity_diffs": None}
    if not args.no_cpu_baseline and world == 1:          # the CPU leg and the parity check: rank 0 of a one-GPU run only
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        from parity_util import count_record_diffs
        orc = oracle_lib.load()
        cores = os.cpu_count() or 1
        sample = min(args.cpu_sample_pairs, batches[args.warmup].shape[0])
        host_bcl = batches[args.warmup][:sample].cpu().numpy()
        ref = orc.reference([c_.cpu().numpy().tobytes() for c_ in genome.contigs])
        ref.set_index(al.get_index())
        p = orc.default_params(2, L, L)
        find_threads = min(cores, 64)      # as MatchFinder: one mask of the table per thread at a time (64 masks)
        # the lookup both ways: entry by entry through the table between two seed k-mers -- the reference's merge join, made for its
        # batches of millions of clusters, which streams all of the table for this sample -- and by bisection of what is left of it; same matches
        orc.lib.oracle_set_lookup_mode(1)
        tc = time.perf_counter()
        om_bisect, _ = ref.find_matches(p, host_bcl, sample, tile=tile_of(0), n_threads=find_threads)
        t_find_bisect = time.perf_counter() - tc
        orc.lib.oracle_set_lookup_mode(0)
        tc = time.perf_counter()
        om, ohits = ref.find_matches(p, host_bcl, sample, tile=tile_of(0), n_threads=find_threads)
        t_find_merge = time.perf_counter() - tc
        assert om_bisect.tobytes() == om.tobytes()
        del om_bisect
        t_find = min(t_find_merge, t_find_bisect)
        otls = oracle_lib.Tls()
        for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
            setattr(otls, name, getattr(tls, name))
        otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
        # (an empty call first: the oracle then has its list of loaded contigs, a copy of the genome, as the reference has its contigs in memory before
        # the first tile; the threads of the timed call take the match list in pieces as they become free)
        ref.select(p, host_bcl, om[:0], otls, all_hits, tile=tile_of(0), n_threads=1, n_clusters_hint=1)
        # on every hardware thread and on half of them (one per core where the host has two a core): the faster run counts
        select_runs = {}
        for n_threads in sorted({cores, max(1, cores // 2)}, reverse=True):
            tc = time.perf_counter()
            orec, ocig, _ = ref.select(p, host_bcl, om, otls, all_hits, tile=tile_of(0), n_threads=n_threads, n_clusters_hint=sample)
            select_runs[n_threads] = time.perf_counter() - tc
        select_threads = min(select_runs, key=select_runs.get)
        t_select = select_runs[select_threads]
        cpu = {"value": round(2.0 * sample / (t_find + t_select), 1), "unit": "reads/s", "cores": select_threads, "kind": "port",
               "sample": "the first %d pairs of the first timed batch; oracle/ (CPU restatement of the reference path, g++ -O3 -mavx2 -ffp-contract=off): seed lookup against the %d-entry "
                         "table on %d threads (merge join %.2f s, bisection %.2f s: the faster one counts) + match selection (%s: the faster one counts; the match list handed to the "
                         "threads in pieces; its banded Smith-Waterman is scalar where the reference's is 16-lane SSE2)"
                         % (sample, n_index, find_threads, t_find_merge, t_find_bisect, ", ".join("%d threads %.2f s" % (k, v) for k, v in sorted(select_runs.items()))),
               "lookup_merge_join_s": round(t_find_merge, 3), "lookup_bisection_s": round(t_find_bisect, 3), "selection_s": round(t_select, 3), "selection_threads": select_threads,
               "selection_runs_s": {str(k): round(v, 3) for k, v in select_runs.items()},
               "value_with_merge_join": round(2.0 * sample / (t_find_merge + t_select), 1), "value_with_bisection": round(2.0 * sample / (t_find_bisect + t_select), 1)}
        # the GPU records of the same pairs (first timed step) against the oracle's, field for field + CIGARs
        grec = checked_records[:2 * sample].cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1)
        gcig = checked_cigars.cpu().numpy().view(np.uint32)
        n_diff, text = count_record_diffs(orec, ocig, grec, gcig)
        parity = {"parity_checked_pairs": int(sample), "parity_diffs": int(n_diff)}
        if text:
            parity["first_diffs"] = text[:3]
        # Every cluster of every step whose MAPQ arithmetic sat within 1e-11 of an integer -- the only place where the device's log10 / exp and
        # glibc's could round a score differently -- re-derived by the oracle (glibc) from its BCL bytes and compared with the GPU's records
        n_flagged = sum(len(b_) for _, b_, _ in flagged)
        parity.update({"mapq_near_integer_pairs": int(n_flagged), "mapq_near_integer_checked": 0, "mapq_near_integer_diffs": 0 if not n_flagged else None})
        if n_flagged:
            f_bcl = np.ascontiguousarray(np.concatenate([b_ for _, b_, _ in flagged]))
            fm, _ = ref.find_matches(p, f_bcl, len(f_bcl), tile=0, n_threads=find_threads)
            frec, fcig, _ = ref.select(p, f_bcl, fm, otls, all_hits, tile=0, n_threads=min(cores, 16), n_clusters_hint=len(f_bcl))
            pools, g_parts, base = [], [], 0
            for s_, _, r_ in flagged:
                r_ = r_.view(abi.FRAGMENT_DTYPE).reshape(-1).copy()
                r_["cigar_offset"] += base
                g_parts.append(r_); pools.append(flagged_cigars[s_]); base += len(flagged_cigars[s_])
            f_diff, f_text = count_record_diffs(frec, fcig, np.concatenate(g_parts), np.concatenate(pools), ignore=("tile", "cluster_id"))
            parity.update({"mapq_near_integer_checked": int(n_flagged), "mapq_near_integer_diffs": int(f_diff)})
            if f_text:
                parity["mapq_first_diffs"] = f_text[:3]
        if bam_info is not None:
            # the BAM record stream of the same pairs: GPU path on its own records against oracle/bam.cpp on the oracle's records, byte for byte
            prefix = "SYNTH:1:%d:" % tile_of(0)
            gbam = al.bam_records([(batches[args.warmup][:sample], checked_records[:2 * sample], checked_cigars, prefix)])[0].cpu().numpy().tobytes()
            obam = orc.bam_records([(host_bcl, orec, ocig, prefix)], [L, L], forced_dodgy_alignment_score=p.dodgy_alignment_score & 0xff)[0]
            bam_info.update({"parity_checked_bytes": len(obam), "parity_identical": gbam == obam})
            if gbam != obam:
                m_ = min(len(gbam), len(obam))
                d_ = np.flatnonzero(np.frombuffer(gbam, np.uint8)[:m_] != np.frombuffer(obam, np.uint8)[:m_])
                bam_info["first_difference_at"] = int(d_[0]) if len(d_) else m_

    workload = "GRCh38-sized synthetic human-like reference (%d bp in %d contigs, 32-mer index %d entries), %d synthetic 2x%d bp pairs per GPU (%d steps x %d pairs)" % (
        args.genome_bases, len(genome), n_index, pairs_rank, L, args.steps, per_rank)
    result = {"metric": "paired-end reads aligned/sec (2x%dbp, GRCh38)" % L, "value": round(reads_per_s, 1), "unit": "reads/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
              "scaling": args.scaling, "vs_baseline": None, "dtype": "u8/int16 (+f64 log-probabilities)", "data": "synthetic",
              "config": {"workload": workload, "pairs_per_step": args.pairs_per_step, "read_length": L, "genome_bases": args.genome_bases, "index_entries": int(n_index),
                         "parallelism": "read shards x%d, %d context(s) per GPU taking the steps' selections in turn (one stream each, contigs and table shared), every step's records and packed CIGARs gathered to rank 0 behind the later steps; %s" % (
                             world, n_contexts, "all lookups before the first selection" if timed_streaming_from is None else
                             "every contig had a match after the lookup of step %d: from there on a step is selected as soon as it is looked up, the later lookups (a context of their own) beside the selections" % timed_streaming_from),
                         "selections_began_after_lookup_of_step": timed_streaming_from, "lookup_phase_ms_per_step": round(1e3 * t_lookups_done / args.steps, 3), "lookup_contexts": len(finders), "hbm_used_gb": hbm_used_gb, "setup_s": round(t_setup, 1),
                         "genome_s": round(t_genome, 1), "index_build_s": round(t_index, 1), "tls": list(tls.astuple()), "pcie_inclusive": pcie, "bam_output": bam_info, "cli_end_to_end": cli_info},
              "roofline": roofline, "cpu_baseline": cpu, "counters": {k: int(v) for k, v in counters.items()}}
    result.update(parity)
    import hashlib
    result["records_sha1"] = hashlib.sha1(checked_records.cpu().numpy().tobytes() + checked_cigars.cpu().numpy().tobytes()).hexdigest()   # first timed step of rank 0
    if gather_identical is not None:
        result["gather_identical"] = gather_identical
    emit(result)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
