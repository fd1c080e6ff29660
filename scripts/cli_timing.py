#!/usr/bin/env python3
"""Times the two programs end to end: a human-like genome written as FASTA -> isaac-sort-reference -> synthetic pairs written as two FASTQ
files (plain) -> isaac-align with the reference's defaults.  Prints the stage lines of both programs and one JSON line (cli_end_to_end).
CLI_GENOME_BASES (1e8), CLI_PAIRS (1e6), CLI_READ_LENGTH (101), CLI_WORK (a directory to work in: /dev/shm/... keeps the 47 GB of mask files
of a GRCh38-sized reference off the disk), CLI_ARGS (more isaac-align options, e.g. "--clusters-at-a-time 4000000 --devices 0,0"; several sets separated
by ";" run isaac-align once each)"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isaac_aligner_amd import build, synth

def main():
    n_bases, n_pairs, L = int(float(os.environ.get("CLI_GENOME_BASES", "1e8"))), int(float(os.environ.get("CLI_PAIRS", "1e6"))), int(os.environ.get("CLI_READ_LENGTH", "101"))
    work = tempfile.mkdtemp(prefix="isaac_cli_", dir=os.environ.get("CLI_WORK") or None)
    dev = torch.device("cuda", 0)
    genome = synth.make_human_like_genome(n_bases, seed=3, device=dev)
    fasta = os.path.join(work, "genome.fa")
    with open(fasta, "wb") as f:
        for i, c in enumerate(genome.contigs if hasattr(genome, "contigs") else genome):
            seq = c.cpu().numpy()
            f.write(b">chr%d synthetic\n" % (i + 1))
            full = len(seq) // 60 * 60
            f.write(np.concatenate([seq[:full].reshape(-1, 60), np.full((full // 60, 1), 10, np.uint8)], axis=1).tobytes())
            if len(seq) > full:
                f.write(seq[full:].tobytes() + b"\n")
    bcl = torch.cat([synth.make_read_pairs(genome, min(1_000_000, n_pairs - first), L, seed=11 + first, device=dev, avoid_gaps=True)[0].cpu() for first in range(0, n_pairs, 1_000_000)]).numpy()
    del genome
    torch.cuda.empty_cache()
    calls = os.path.join(work, "calls"); os.makedirs(calls)
    t0 = time.time()
    for read in range(2):
        b = bcl[:, read * L:(read + 1) * L]
        is_n = (b & 0xFC) == 0
        bases = np.where(is_n, ord("N"), np.frombuffer(b"ACGT", np.uint8)[b & 3]).astype(np.uint8)
        quals = np.where(is_n, 35, 33 + (b >> 2)).astype(np.uint8)
        header = np.frombuffer(b"@M1:7:FCBENCH:1:1101:", np.uint8)
        rows = []
        digits = np.char.zfill(np.arange(n_pairs).astype(str), 8).astype("S8").view(np.uint8).reshape(n_pairs, 8)
        nl = np.full((n_pairs, 1), ord("\n"), np.uint8); plus = np.full((n_pairs, 1), ord("+"), np.uint8)
        text = np.concatenate([np.tile(header, (n_pairs, 1)), digits, nl, bases, nl, plus, nl, quals, nl], axis=1)
        text.tofile(os.path.join(calls, "lane1_read%d.fastq" % (read + 1)))
    print("FASTQ written in %.1f s (%d MB)" % (time.time() - t0, 2 * n_pairs * (21 + 8 + 4 + 2 * L) // 1000000), flush=True)
    ref_dir = os.path.join(work, "ref")
    tools = os.path.dirname(build.build_host())
    # CLI_ARGS may hold several option sets separated by ';': isaac-align runs once for each, on the same files
    runs = [("isaac-sort-reference", [os.path.join(tools, "isaac-sort-reference"), "-g", fasta, "-o", ref_dir, "-q"])]
    envs = {}
    for extra in os.environ.get("CLI_ARGS", "").split(";"):
        tokens = extra.split()
        env = dict(t.split("=", 1) for t in tokens if "=" in t and t.split("=", 1)[0].isupper())          # NAME=value tokens: environment of that run
        runs.append(("isaac-align", [os.path.join(tools, "isaac-align"), "-r", os.path.join(ref_dir, "sorted-reference.xml"), "-b", calls, "--base-calls-format", "fastq", "-o", os.path.join(work, "Aligned")]
                     + [t for t in tokens if not ("=" in t and t.split("=", 1)[0].isupper())]))
        envs[len(runs) - 1] = env
    for k, (name, cmd) in enumerate(runs):
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **envs.get(k, {})))
        wall = time.time() - t0
        print("%s: rc %d, %.1f s%s" % (name, r.returncode, wall, "  [" + " ".join(cmd[9:] + ["%s=%s" % kv for kv in envs.get(k, {}).items()]) + "]" if name == "isaac-align" else ""))
        print("\n".join(l for l in r.stderr.splitlines() if "done in" in l or "records" in l or "clusters in" in l or "error" in l.lower() or "timing" in l), flush=True)
        if name == "isaac-align" and r.returncode == 0:
            import json
            timing = json.loads([l for l in r.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
            timing.update({"wall_s": round(wall, 2), "reads_per_s": round(timing["reads"] / wall, 1), "reads_per_s_without_reference_load": round(timing["reads"] / (timing["total_s"] - timing["reference_s"]), 1),
                           "genome_bases": n_bases, "read_length": L})
            print("cli_end_to_end " + json.dumps(timing), flush=True)
            import hashlib
            for suffix in ("", ".bai"):
                h = hashlib.md5()
                with open(os.path.join(work, "Aligned", "Projects", "default", "default", "sorted.bam" + suffix), "rb") as f:
                    for block in iter(lambda: f.read(1 << 26), b""):
                        h.update(block)
                print("md5 sorted.bam%s %s" % (suffix, h.hexdigest()), flush=True)
            if os.environ.get("CLI_INFLATE"):
                # the file's content against the previous run's: the records must not depend on the options that only deal the work out
                import zlib
                from isaac_aligner_amd import bam as bam_host
                raw = open(os.path.join(work, "Aligned", "Projects", "default", "default", "sorted.bam"), "rb").read()
                parts, at = [], 0
                while at < len(raw):
                    size = int.from_bytes(raw[at + 16:at + 18], "little") + 1
                    parts.append(zlib.decompress(raw[at + 18:at + size - 8], -15))
                    at += size
                content = np.frombuffer(b"".join(parts), np.uint8)
                del raw, parts
                global previous_content
                def records_of(arr):                      # behind the header (its @PG line carries the command line, which differs between the runs)
                    l_text = int.from_bytes(arr[4:8].tobytes(), "little"); p = 8 + l_text
                    n_ref = int.from_bytes(arr[p:p + 4].tobytes(), "little"); p += 4
                    for _ in range(n_ref):
                        l_name = int.from_bytes(arr[p:p + 4].tobytes(), "little"); p += 8 + l_name
                    return arr[p:]
                content = records_of(content)
                if "previous_content" in globals() and previous_content is not None:
                    same = len(content) == len(previous_content) and bool((content == previous_content).all())
                    print("inflated records %d bytes, identical to the previous run's: %s" % (len(content), same), flush=True)
                    if not same:
                        n = min(len(content), len(previous_content))
                        first = int(np.argmax(content[:n] != previous_content[:n]))
                        print("first difference at byte %d of %d / %d" % (first, len(content), len(previous_content)))
                        p = 0
                        while p < n:
                            size = int.from_bytes(previous_content[p:p + 4].tobytes(), "little")
                            if p + 4 + size > first:
                                break
                            p += 4 + size
                        for arr, label in ((previous_content, "previous"), (content, "this")):
                            r = bam_host.parse_records(arr[p:p + 4 + int.from_bytes(arr[p:p + 4].tobytes(), "little")].tobytes())[0]
                            r.pop("qual"); r.pop("seq")
                            print(label, r)
                previous_content = content
    bam = os.path.join(work, "Aligned", "Projects", "default", "default", "sorted.bam")
    print("sorted.bam %d MB, .bai %d KB" % (os.path.getsize(bam) // 1000000, os.path.getsize(bam + ".bai") // 1000))
    subprocess.run(["rm", "-rf", work])

main()
