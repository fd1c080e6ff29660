#!/usr/bin/env python3
"""The headline configuration through the drop-in itself: bin/isaac-align on the GRCh38-sized synthetic reference (3.1 Gbp, 25 contigs, 2.93 G table entries read
back from 64 mask files) and N million synthetic 2x150 pairs in FASTQ lanes, with the reference's defaults (duplicates marked, gaps realigned, --bam-gzip-level 1).

    python scripts/cli_headline.py --pairs 100000000 --lanes 4 [--devices 0,0] [--out profiles/r5_cli_headline.json]

Files go to /dev/shm (the box has 1.5 TB there).  Reports pairs, wall time, the program's own stage timers, peak device and host memory, and compares the
records of three sampled tiles -- dumped by the program as it selected them (ISAAC_ALIGN_DUMP_TILES) -- with the oracle run on the same tiles' base calls
(seed lookup against the same table, the lane's template-length statistics from its first tile, selection), record for record and CIGAR word for CIGAR word.
The oracle is the checker only; everything timed is the product."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100_000_000)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--genome-bases", type=int, default=3_100_000_000)
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--clusters-at-a-time", type=int, default=4_000_000)
    ap.add_argument("--sample-tiles", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r5_cli_headline.json"))
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--extra-env", default="", help="NAME=value[,NAME=value][|more options]: the program once more on the same files with these set (timing only); several sets separated by ';'")
    args = ap.parse_args()
    import numpy as np
    import torch
    from isaac_aligner_amd import abi, build, gpu, options, sorted_reference as sr, synth
    L = args.read_length
    dev = torch.device("cuda", 0)
    work = tempfile.mkdtemp(prefix="isaac_headline_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    result = {"pairs": args.pairs, "lanes": args.lanes, "read_length": L, "genome_bases": args.genome_bases, "devices": args.devices, "clusters_at_a_time": args.clusters_at_a_time}
    try:
        t0 = time.time()
        genome = synth.make_human_like_genome(args.genome_bases, seed=3, device=dev)
        params = options.default_params(L, L)
        al = gpu.Aligner(params, 0, genome)
        n_index = al.build_index(repeat_threshold=1000)
        result["index_entries"] = int(n_index); result["index_build_s"] = round(time.time() - t0, 1)
        ref_dir, calls = os.path.join(work, "ref"), os.path.join(work, "calls")
        os.makedirs(ref_dir); os.makedirs(calls)
        fasta = os.path.join(ref_dir, "genome.fa")
        t0 = time.time()
        contigs, position = [], 0
        for i, (offset, size, bases, acgt) in enumerate(synth.write_fasta(fasta, genome.contigs)):
            m = sr.Contig()
            m.genomic_position, m.index, m.karyotype_index, m.name, m.file = position, i, i, b"chr%d" % (i + 1), fasta.encode()
            m.offset, m.size, m.total_bases, m.acgt_bases = offset, size, bases, acgt
            position += bases
            contigs.append(m)
        al.save_sorted_reference(ref_dir, "genome.fa", contigs)
        result["reference_files_s"] = round(time.time() - t0, 1)
        # the reads: batches of 1 M pairs drawn on the device, written as FASTQ text by a thread per lane (numpy releases the interpreter lock in the big operations)
        t0 = time.time()
        per_batch = 1_000_000
        n_batches = (args.pairs + per_batch - 1) // per_batch
        lane_of = lambda b: b % args.lanes
        files = [[open(os.path.join(calls, "lane%d_read%d.fastq" % (lane + 1, r + 1)), "wb") for r in range(2)] for lane in range(args.lanes)]
        written = [0] * args.lanes
        pool = ThreadPoolExecutor(max_workers=args.lanes)
        pending = [None] * args.lanes

        def write(lane, bcl, first):
            synth.write_fastq(files[lane], bcl, L, name_prefix=b"M1:7:FCHEAD:%d:1101:" % (lane + 1), first_index=first)
        kept = {}                                       # base calls of the batches, by (lane, first cluster): the oracle's input for the sampled tiles is cut from the FASTQ text instead
        for b in range(n_batches):
            n = min(per_batch, args.pairs - b * per_batch)
            bcl = synth.make_read_pairs(genome, n, L, seed=5000 + b, device=dev, avoid_gaps=True)[0].cpu().numpy()
            lane = lane_of(b)
            if pending[lane] is not None:
                pending[lane].result()
            pending[lane] = pool.submit(write, lane, bcl, written[lane])
            written[lane] += n
        for p in pending:
            if p is not None:
                p.result()
        pool.shutdown()
        for fs in files:
            for f in fs:
                f.close()
        result["fastq_files_s"] = round(time.time() - t0, 1)
        result["fastq_bytes"] = sum(os.path.getsize(os.path.join(calls, f)) for f in os.listdir(calls))
        del al
        del genome
        torch.cuda.empty_cache()
        # the tiles to sample: the program numbers its tiles over the run in lane order; a lane of c clusters has ceil(c / tile) tiles
        tile = args.clusters_at_a_time
        def tiles_of_lane(clusters):                    # FastqSeedSource: loads of --clusters-at-a-time clusters, each cut into tiles (isaac_gpu_fastq_tiles)
            n, left = 0, clusters
            while left > 0:
                load = min(left, tile)
                n += len(gpu.fastq_tiles(load, params.n_seeds, clusters_at_a_time=tile)[0])
                left -= load
            return n
        tiles_per_lane = [tiles_of_lane(w) for w in written]
        rng = np.random.default_rng(7)
        everything = [(lane + 1, number + 1) for lane, k in enumerate(tiles_per_lane) for number in range(k)]       # (lane number, tile number in the lane): as in the read names
        sample = set(everything[int(x)] for x in rng.choice(len(everything), size=min(args.sample_tiles, len(everything)), replace=False))
        # ... and the first tile of every lane a sampled tile is in: the lane's template-length statistics come from it
        first_of_lane = {lane: (lane, 1) for lane, _ in sample}
        sample = sorted(sample | set(first_of_lane.values()))
        dump = os.path.join(work, "dump")
        tool = build.build_host()
        cmd = [tool, "-r", os.path.join(ref_dir, "sorted-reference.xml"), "-b", calls, "--base-calls-format", "fastq", "-o", os.path.join(work, "Aligned"), "--use-bases-mask", "y*,y*",
               "--clusters-at-a-time", str(tile), "--devices", args.devices]
        env = dict(os.environ, ISAAC_ALIGN_DUMP_TILES="%s:%s" % (dump, ",".join("%d.%d" % s for s in sample)))
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        wall = time.time() - t0
        result["command"] = "isaac-align -r sorted-reference.xml -b <%d lanes x 2 FASTQ files> --base-calls-format fastq --use-bases-mask y*,y* --clusters-at-a-time %d --devices %s" % (args.lanes, tile, args.devices)
        result["rc"] = r.returncode; result["wall_s"] = round(wall, 2)
        result["stderr_tail"] = r.stderr[-1500:]
        if r.returncode:
            json.dump(result, open(args.out, "w"), indent=1)
            print(json.dumps(result)[:3000])
            return 1
        timing = json.loads([l for l in r.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
        result["timing"] = timing
        result["reads_per_s"] = round(timing["reads"] / wall, 1)
        result["reads_per_s_without_reference_load"] = round(timing["reads"] / max(1e-9, timing["total_s"] - timing["reference_s"]), 1)
        bam_path = os.path.join(work, "Aligned", "Projects", "default", "default", "sorted.bam")
        result["sorted_bam_bytes"] = os.path.getsize(bam_path); result["bai_bytes"] = os.path.getsize(bam_path + ".bai")
        result["peak_device_gb"] = round(timing["peak_device_bytes"] / 1e9, 1); result["peak_host_gb"] = round(timing["peak_host_bytes"] / 1e9, 1)
        # ---- the same files again with other switches: the stage timers only (the first run's output is what is checked)
        result["extra_runs"] = []
        for extra in [e for e in args.extra_env.split(";") if e.strip()]:
            shutil.rmtree(os.path.join(work, "Aligned"), ignore_errors=True)
            env_part, _, arg_part = extra.partition("|")             # NAME=value,... | more options (a later option replaces an earlier one)
            more = dict(kv.split("=", 1) for kv in env_part.split(",") if "=" in kv)
            t0 = time.time()
            r2 = subprocess.run(cmd + arg_part.split(), capture_output=True, text=True, env=dict(os.environ, **more))
            w2 = time.time() - t0
            entry = {"env": more, "args": arg_part.strip(), "rc": r2.returncode, "wall_s": round(w2, 2)}
            if not r2.returncode:
                t2 = json.loads([l for l in r2.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
                t2.pop("bin_ranges", None)
                entry["timing"] = t2
                entry["reads_per_s"] = round(t2["reads"] / w2, 1)
                entry["reads_per_s_without_reference_load"] = round(t2["reads"] / max(1e-9, t2["total_s"] - t2["reference_s"]), 1)
                entry["sorted_bam_bytes"] = os.path.getsize(bam_path)
            else:
                entry["stderr_tail"] = r2.stderr[-800:]
            result["extra_runs"].append(entry)
        # ---- the sampled tiles against the oracle
        import oracle_lib
        from parity_util import compare_records
        o = oracle_lib.load()
        t0 = time.time()
        table = np.concatenate([np.fromfile(os.path.join(ref_dir, f), abi.REFERENCE_KMER_DTYPE) for f in sorted(f for f in os.listdir(ref_dir) if f.endswith(".dat"))])
        fa = open(fasta, "rb").read()
        host_contigs = [bytes(c for c in fa[m.offset:m.offset + m.size] if c != 10) for m in contigs]
        del fa
        ref = o.reference(host_contigs)
        ref.set_index(table)
        all_hits = np.ones(len(host_contigs), np.uint8)
        checks, tls_checks = [], []
        for s in sample:
            stem = os.path.join(dump, "tile_%d_%d" % s)
            meta = json.load(open(stem + ".json"))
            n = meta["clusters"]
            bcl = np.fromfile(stem + ".bcl", np.uint8).reshape(n, 2 * L)
            rec = np.fromfile(stem + ".records", abi.FRAGMENT_DTYPE)
            cig = np.fromfile(stem + ".cigars", np.uint32)
            tls = abi.Tls()
            v = meta["tls"]
            tls.min, tls.max, tls.median, tls.low_std_dev, tls.high_std_dev = v[0:5]
            tls.best_model[0], tls.best_model[1], tls.stable, tls.mate_min, tls.mate_max = v[5:10]
            threads = min(128, os.cpu_count() or 1)
            om, hits = ref.find_matches(params, bcl, n, tile=meta["index"] & 0xfff, n_threads=threads)
            if s in first_of_lane.values():
                # the lane's statistics as the oracle learns them from the lane's first tile: what the program used for every tile of the lane
                otls = ref.determine_tls(params, bcl, om, all_hits, tile=meta["index"])
                tls_checks.append({"tile": "%d.%d" % s, "lane": meta["lane"], "oracle": list(otls.astuple()), "program": list(tls.astuple()), "equal": otls.astuple() == tls.astuple()})
            orec, ocig, _ = ref.select(params, bcl, om, tls, all_hits, tile=meta["index"], n_threads=threads, n_clusters_hint=n)
            # every field of every record, and every CIGAR word through each record's own offset (the oracle's CIGARs lie in slots, the program's are packed)
            bad = np.zeros(len(rec), bool) if len(rec) == len(orec) else np.ones(max(len(rec), len(orec)), bool)
            if len(rec) == len(orec):
                for f in rec.dtype.names:
                    if f not in ("cigar_offset", "reserved"):
                        bad |= rec[f] != orec[f]
                bad |= (rec["reserved"] >> 16) != (orec["reserved"] >> 16)
                lengths = rec["cigar_length"].astype(np.int64)
                starts = np.cumsum(lengths) - lengths
                within = np.arange(int(lengths.sum()), dtype=np.int64) - np.repeat(starts, lengths)
                mine = cig[np.repeat(rec["cigar_offset"].astype(np.int64), lengths) + within]
                theirs = ocig[np.repeat(orec["cigar_offset"].astype(np.int64), lengths) + within]
                word_bad = mine != theirs
                if word_bad.any():
                    bad[np.unique(np.repeat(np.arange(len(rec)), lengths)[word_bad])] = True
            first = [int(i) for i in np.flatnonzero(bad)[:3]]
            checks.append({"tile": "%d.%d" % s, "lane": meta["lane"], "clusters": n, "records": int(len(rec)), "diffs": int(bad.sum()),
                           "first_diffs": compare_records(orec[first], ocig, rec[first], cig) if first and len(rec) == len(orec) else []})
        result["sampled_tiles"] = checks
        result["lane_statistics"] = tls_checks
        result["sampled_parity_diffs"] = sum(c["diffs"] for c in checks) + sum(0 if c["equal"] else 1 for c in tls_checks)
        result["oracle_check_s"] = round(time.time() - t0, 1)
        json.dump(result, open(args.out, "w"), indent=1)
        print(json.dumps({k: v for k, v in result.items() if k != "stderr_tail"}))
        return 0 if not result["sampled_parity_diffs"] else 2
    finally:
        if not args.keep:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
