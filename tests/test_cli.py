"""isaac-align, the command-line host (isaac_aligner_amd/host, built on include/isaac_gpu.h alone), and the host-only entry points it
brought: option defaults / --seeds / --gap-scoring (isaac_gpu_default_params, isaac_gpu_parse_seeds, isaac_gpu_parse_gap_scoring) and the
.bai writer (isaac_gpu_bam_index) against the oracle's restatement of bam::BamIndexPart / bam::BamIndex and against what a BAI must mean.
On the GPU: a two-lane FASTQ flowcell through the binary, sorted.bam and sorted.bam.bai compared byte for byte with the oracle run on the
same inputs (reader, seed lookup, template statistics per lane, selection, duplicate marking, gap realignment, record stream, index)."""
import ctypes as C
import gzip
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib
from isaac_aligner_amd import abi, bam, build, gpu, options, sorted_reference as sr, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host():
    return build.build_host()


def run_host(*args, cwd=None, env=None):
    return subprocess.run([host()] + [str(a) for a in args], capture_output=True, text=True, cwd=cwd, env=dict(os.environ, **env) if env else None)


# ---- options ---------------------------------------------------------------------------------------------------------------------------

PARAM_FIELDS = ("gap_match", "gap_mismatch", "gap_open", "gap_extend", "min_gap_extend", "repeat_threshold", "gapped_mismatches_max", "semialigned_gap_limit", "base_quality_cutoff",
                "ignore_neighbors", "clip_semialigned", "clip_overlapping", "scatter_repeats", "dodgy_alignment_score", "mapq_threshold", "keep_unaligned", "mate_drift_range",
                "first_pass_seeds", "seed_length", "n_reads", "n_seeds")


def seeds_of(p):
    return [(p.seeds[i].offset, p.seeds[i].length, p.seeds[i].read_index) for i in range(p.n_seeds)]


@pytest.mark.parametrize("lengths", [(100, 100), (150, 150), (36, 0), (50, 50), (64, 64), (33, 33), (250, 250), (101, 75), (32, 32)])
def test_default_params_follow_the_options_rule(lengths):
    """isaac_gpu_default_params (C, what the command-line host starts from) == options.default_params (Python) == the oracle's"""
    lib = gpu.load_library()
    p = abi.Params()
    assert lib.isaac_gpu_default_params(C.c_uint32(lengths[0]), C.c_uint32(lengths[1]), C.byref(p)) == 0
    want = options.default_params(*lengths)
    assert [getattr(p, f) for f in PARAM_FIELDS] == [getattr(want, f) for f in PARAM_FIELDS]
    assert seeds_of(p) == seeds_of(want) and list(p.read_length) == list(want.read_length)
    o = oracle_lib.load().default_params(2 if lengths[1] else 1, *lengths)
    assert seeds_of(p) == seeds_of(o) and p.first_pass_seeds == o.first_pass_seeds


def test_seed_descriptors_and_gap_scoring():
    lib = gpu.load_library()
    lib.isaac_gpu_params_last_error.restype = C.c_char_p
    p = options.default_params(100, 100)
    # SeedDescriptorOption.cpp: a list per read, the last one serves the reads that follow; seeds that do not fit are dropped
    assert lib.isaac_gpu_parse_seeds(b"0:32:64:90", C.c_uint32(3), C.byref(p)) == 0
    assert seeds_of(p) == [(0, 32, 0), (32, 32, 0), (64, 32, 0), (0, 32, 1), (32, 32, 1), (64, 32, 1)] and p.first_pass_seeds == 3
    assert lib.isaac_gpu_parse_seeds(b"0:16,8", C.c_uint32(2), C.byref(p)) == 0
    assert seeds_of(p) == [(0, 32, 0), (16, 32, 0), (8, 32, 1)] and p.first_pass_seeds == 1
    p.semialigned_gap_limit = 0
    assert lib.isaac_gpu_parse_seeds(b"auto", C.c_uint32(1), C.byref(p)) == 0 and p.first_pass_seeds == 1       # 2 only with a gap limit (AlignOptions.cpp:1165-1171)
    for bad, message in ((b"", b"empty"), (b"0:x", b"Invalid seed offset 'x'"), (b"0,0,0", b"Too many lists-of-seeds"), (b"90", b"At least one seed must be used")):
        assert lib.isaac_gpu_parse_seeds(bad, C.c_uint32(1), C.byref(p)) == 1 and message in lib.isaac_gpu_params_last_error()
    assert lib.isaac_gpu_parse_gap_scoring(b"eland", C.byref(p)) == 0 and (p.gap_match, p.gap_mismatch, p.gap_open, p.gap_extend, p.min_gap_extend) == (2, -1, -15, -3, -25)
    assert lib.isaac_gpu_parse_gap_scoring(b"1:-2:-3:-4:-5", C.byref(p)) == 0 and (p.gap_match, p.gap_mismatch, p.gap_open, p.gap_extend, p.min_gap_extend) == (1, -2, -3, -4, -5)
    for bad, message in ((b"1:2", b"five components"), (b"-1:-2:-3:-4:-5", b"match score"), (b"1:2:-3:-4:-5", b"mismatch score"), (b"1:-2:-3:-4:5", b"score cap")):
        assert lib.isaac_gpu_parse_gap_scoring(bad, C.byref(p)) == 1 and message in lib.isaac_gpu_params_last_error()


def test_default_adapters_syntax():
    """isaac_gpu_parse_adapters: flowcell::SequencingAdapterListGrammar + parseDefaultAdapters (include/flowcell/SequencingAdapterListGrammar.hpp:52-104,
    lib/options/alignOptions/DefaultAdaptersOption.cpp:35-60) -- the three macros with the reference's own lists (tests/golden/sequencing_adapter.json holds them as
    data), plain / forward-unbounded / reverse-unbounded sequences, lower case, optional commas, and the error text for what is left unparsed"""
    import json
    lib = gpu.load_library()
    lib.isaac_gpu_params_last_error.restype = C.c_char_p
    presets = json.load(open(os.path.join(ROOT, "tests", "golden", "sequencing_adapter.json")))["presets"]
    p = options.default_params(100, 100)

    def parsed(text):
        assert lib.isaac_gpu_parse_adapters(text.encode(), C.byref(p)) == 0, lib.isaac_gpu_params_last_error()
        return [(p.adapters[i].sequence.decode(), bool(p.adapters[i].reverse), p.adapters[i].clip_length) for i in range(p.n_adapters)]
    for macro, name in (("Standard", "STANDARD_ADAPTERS"), ("Nextera", "NEXTERA_STANDARD_ADAPTERS"), ("NexteraMp", "NEXTERA_MATEPAIR_ADAPTERS")):
        want = [(a["sequence"], a["reverse"], a["clip_length"]) for a in presets[name]]
        assert parsed(macro) == want, macro
        assert [tuple(x) for x in options.ADAPTER_PRESETS[macro]] == want, macro
    assert parsed("ACGTA*,*TGCAT") == [("ACGTA", False, 0), ("TGCAT", True, 0)]
    assert parsed("acgtac,TGCATG") == [("ACGTAC", False, 6), ("TGCATG", False, 6)]
    assert parsed("ACGTA*TGCATGG*ACGTT") == [("ACGTA", False, 0), ("TGCATGG", False, 0), ("ACGTT", False, 5)]      # the comma is optional
    assert parsed("") == []
    for bad, at in ((b"ACGT", b"at: ACGT "), (b"ACGTA*,x", b"at: x "), (b"Nextera,ACGTA", b"at: ,ACGTA "), (b"*ACGTA*", b"at: * "), (b"ACGTN", b"at: ACGTN ")):
        assert lib.isaac_gpu_parse_adapters(bad, C.byref(p)) == 1 and b"Could not parse the default-adapters" in lib.isaac_gpu_params_last_error() and at in lib.isaac_gpu_params_last_error(), bad
    assert lib.isaac_gpu_parse_adapters(b"A" * 127, C.byref(p)) == 1 and b"too long" in lib.isaac_gpu_params_last_error()


def test_command_line_errors(tmp_path):
    """what options::AlignOptions rejects, and what this host refuses instead of ignoring; exit code 1 with the message (common::run)"""
    assert run_host("--version").stdout.strip().startswith("isaac_aligner_amd")
    assert "--base-calls-format" in run_host("--help").stdout
    cases = [(["-r", "x.xml"], "At least one 'base-calls' is required"),
             (["-b", tmp_path], "At least one 'reference-genome' is required"),
             (["-r", "x.xml", "-b", tmp_path], "this host reads fastq and fastq-gz only"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--frobnicate", "1"], "unrecognised option '--frobnicate'"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--realign-gaps", "maybe"], "The 'realign-gaps' value is invalid maybe"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--keep-unaligned", "sideways"], "must be 'discard', 'front' or 'back'"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--dodgy-alignment-score", "300"], "must be either Unknown, Unaligned or a number 0-255 (300 given)"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--seed-length", "20"], "--seed-length other than 16, 32 or 64 is not supported"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--seed-length", "64"], "32-mer seeds only"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--avoid-smith-waterman", "1"], "--avoid-smith-waterman 1 is not supported"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--mark-duplicates", "perhaps"], "option '--mark-duplicates' is invalid"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "--jobs"], "the required argument for option '--jobs' is missing"),
             (["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq", "-m", "5", "-j4", "-t", tmp_path / "Temp"], "Could not find any fastq lanes in")]
    for args, message in cases:
        r = run_host(*args)
        assert r.returncode == 1 and message in r.stderr, (args, r.stderr)
    # a flowcell that is there: ids of the two reads must agree, masks must parse and leave a prefix of every read
    for lane, (h1, h2) in enumerate([("@M1:7:FCA:1:1", "@M1:7:FCB:1:1")], start=1):
        for read, header in ((1, h1), (2, h2)):
            (tmp_path / ("lane%d_read%d.fastq" % (lane, read))).write_text("%s\n%s\n+\n%s\n" % (header, "ACGT" * 10, "I" * 40))
    base = ["-r", "x.xml", "-b", tmp_path, "--base-calls-format", "fastq"]
    r = run_host(*base)
    assert r.returncode == 1 and "Flowcell ID mismatch between fastq reads FCA vs FCB" in r.stderr
    (tmp_path / "lane1_read2.fastq").write_text("@M1:7:FCA:1:1\n%s\n+\n%s\n" % ("ACGT" * 10, "I" * 40))
    for mask, message in (("y*n,q*", "Could not parse the use-bases-mask 'y*n,q*'"), ("y*", "incompatible with number of reads (2)"), ("n4y*,y*", "is not of that form"),
                          ("y*n,i*", "index cycles are not supported"), ("y20n*,y*", "is too short: 20 cycle < 32"), ("y0n*,y*", "Could not parse")):
        r = run_host(*(base + ["--use-bases-mask", mask]))
        assert r.returncode == 1 and message in r.stderr, (mask, r.stderr)
    r = run_host(*(base + ["--use-bases-mask", "y36n*,y*n"]))          # a good mask: the run gets as far as the device (or the reference file)
    assert r.returncode == 1 and ("isaac_gpu_create" in r.stderr or "x.xml" in r.stderr)


def test_run_planning_without_a_device(tmp_path):
    """ISAAC_ALIGN_PLAN_ONLY: what isaac-align decides before it touches a device -- the threads that read lanes and their loader contexts, the loads, whether the
    selection is streamed beside the loading, the bins (planBins) -- from sorted-reference.xml and the sizes of the FASTQ files alone"""
    import json
    from isaac_aligner_amd import sorted_reference as sr
    lengths = [50_000_000, 3_000, 700_000, 20_000_000, 5_000]                  # karyotype order below is not the file's order
    karyotype = [3, 0, 4, 1, 2]
    contigs, position = [], 0
    for i, length in enumerate(lengths):
        c = sr.Contig()
        c.genomic_position, c.index, c.karyotype_index, c.name, c.file = position, i, karyotype[i], b"c%d" % i, b"genome.fa"
        c.offset, c.size, c.total_bases, c.acgt_bases = position, length, length, length
        position += length
        contigs.append(c)
    xml = tmp_path / "sorted-reference.xml"
    xml.write_text(sr.format(contigs, []))
    calls = tmp_path / "calls"
    calls.mkdir()
    record = "@M1:7:FCPLAN:%d:1101:00000001\n" + "A" * 100 + "\n+\n" + "I" * 100 + "\n"
    lanes_clusters = {1: 30_000, 2: 90_000, 4: 60_000}
    for lane, n in lanes_clusters.items():
        one = (record % lane).encode()
        for read in (1, 2):
            with open(calls / ("lane%d_read%d.fastq" % (lane, read)), "wb") as f:
                f.write(one * 1000)
                f.truncate(len(one) * n)                                        # a sparse file of the size n such records have: only sizes are read here
    base = ["-r", xml, "-b", calls, "--base-calls-format", "fastq", "-o", tmp_path / "Aligned"]
    ordered = [lengths[i] for i in sorted(range(len(lengths)), key=lambda i: karyotype[i])]
    total = sum(lanes_clusters.values())

    def plan(*more, **env):
        r = run_host(*(base + list(more)), env=dict({"ISAAC_ALIGN_PLAN_ONLY": "1"}, **env))
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout.strip().splitlines()[-1])

    p = plan("--clusters-at-a-time", "10000", "--bin-records", "40000")
    assert p["estimated_clusters"] == total and p["lanes"] == 3 and p["workers"] == 1
    assert p["readers"] == 2 and p["loader_contexts"] == 4                     # one reader more than there are workers; a context per reader and read
    assert p["load_clusters"] == 10000 and p["expected_loads"] == 18 and p["selection_streamed"] == 1        # eight loads per worker and more: streamed
    ranges, cuts = plan_bins(ordered, 40000 / (total * 2 / sum(ordered)))
    assert p["bins"] == len(ranges) + 1 and p["bin_cuts"] == len(cuts) > 0 and [tuple(r) for r in p["bin_ranges"]] == ranges
    # every bin begins where the one before it ends, a contig's first bin at its first base, and no bin spans two long contigs
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and ranges[0][0] == bam.reference_position(0, 0) and ranges[-1][1] == bam.reference_position(len(ordered), 0)
    p = plan("--clusters-at-a-time", "10000", "--devices", "0,0,0")
    assert p["workers"] == 3 and p["readers"] == 3 and p["loader_contexts"] == 6 and p["selection_streamed"] == 0        # 18 loads on three workers: not a long run
    assert p["bins"] == 2 and p["bin_cuts"] == 0                               # 360 000 records fit one bin of 4 M: every contig in it, and the unaligned bin
    assert plan("--clusters-at-a-time", "10000", "--devices", "0,0,0", ISAAC_ALIGN_STREAM_SELECTION="1")["selection_streamed"] == 1
    p = plan()
    assert p["load_clusters"] == 4 * p["tile_clusters_max"] and p["expected_loads"] == 1 and p["selection_streamed"] == 0


def test_sort_reference_command_line(tmp_path):
    """bash/bin/isaac-sort-reference's argument handling (exit codes 1 for help / version, 2 for errors) and the contig table it reads;
    without a GPU it stops at the device with a message"""
    tool = os.path.join(os.path.dirname(host()), "isaac-sort-reference")
    run = lambda *a: subprocess.run([tool] + [str(x) for x in a], capture_output=True, text=True)
    assert run("-v").returncode == 1 and run("-h").returncode == 1 and "--genome-file" in run("-h").stdout
    for args, code, message in ((["--frob"], 2, "ERROR: unrecognized argument: --frob"), ([], 2, "--output-directory and --genome-file arguments are mandatory"),
                                (["-g", tmp_path / "missing.fa"], 2, "ERROR: File not found"), (["-g", __file__, "-s", "20"], 2, "--seed-length must be 16, 32 or 64"),
                                (["-g", __file__, "-s", "64"], 2, "32-mer references only"), (["-g", __file__, "-w", "4"], 2, "--mask-width 6"), (["-g", __file__, "-n"], 2, "--dry-run")):
        r = run(*args)
        assert r.returncode == code and message in r.stderr + r.stdout, (args, r.stderr, r.stdout)
    fasta = tmp_path / "two.fa"
    fasta.write_bytes(b">c1 first contig\nACGTNNACGTacgtRYACGTTGCA\nACGT\n>c2\nGGGGCCCC\n")
    r = run("-g", fasta, "-o", tmp_path / "out")
    if "isaac_gpu_create" in r.stderr:          # no GPU here: the contig table was read, the device was not there
        assert r.returncode == 2
    else:
        assert r.returncode == 0, r.stderr
    assert "contig c1: 28 bases (24 ACGT) at byte 17, M5 c069df0a1d4472f4a9fbeebe6124a937" in r.stderr and "contig c2: 8 bases (8 ACGT) at byte 51, M5 9b2ef89d932478a21dc98f32c1f2346f" in r.stderr


# ---- the index -------------------------------------------------------------------------------------------------------------------------

def bgzf_blocks(data):
    """[(offset, compressed size, uncompressed bytes)] of a run of BGZF blocks"""
    blocks, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 12:at + 14] == b"BC"
        size = int.from_bytes(data[at + 16:at + 18], "little") + 1
        blocks.append((at, size, zlib.decompress(data[at + 18:at + size - 8], -15)))
        assert int.from_bytes(data[at + size - 4:at + size], "little") == len(blocks[-1][2])
        at += size
    return blocks


def check_index_semantics(bai, stream, parts, n_contigs, header_bgzf_bytes, read_length):
    """independent of both implementations: every record with a position is inside a chunk of its bin, the linear index of a 16 kb window is
    the smallest virtual offset of the records that overlap it, the counts are the records' counts"""
    contigs, no_coordinate = bam.parse_index(bai)
    assert len(contigs) == n_contigs
    voffsets, file_at = {}, header_bgzf_bytes
    for offset, n, bgzf in parts:
        starts, raw = [], 0
        for at, size, data in bgzf_blocks(bgzf):
            starts.append((raw, file_at + at)); raw += len(data)
        assert raw == n
        at = 0
        while at < n:
            k = max(i for i, (u, _) in enumerate(starts) if u <= at)
            voffsets[offset + at] = (starts[k][1] << 16) | (at - starts[k][0])
            at += 4 + int.from_bytes(stream[offset + at:offset + at + 4], "little")
        file_at += len(bgzf)
    recs, at = bam.parse_records(stream), 0
    windows = [dict() for _ in range(n_contigs)]
    counts = [[0, 0] for _ in range(n_contigs)]
    unplaced = 0
    for r in recs:
        v = voffsets[at]
        at += 4 + int.from_bytes(stream[at:at + 4], "little")
        if r["pos"] < 0:
            unplaced += 1
            continue
        c = contigs[r["ref_id"]]
        span = sum(int(w) >> 4 for w in r["cigar"] if (int(w) & 15) in (0, 2, 3, 7, 8))
        index_bin = bam_reg2bin(r["pos"], r["pos"] + len(r["seq"]))               # the read's length, not its span: "samtools is doing it this way"
        assert any(b <= v < e for b, e in c["bins"][index_bin]), (r["name"], index_bin)
        for w in {r["pos"] >> 14, (r["pos"] + max(span, 1) - 1) >> 14}:
            windows[r["ref_id"]][w] = min(windows[r["ref_id"]].get(w, v), v)
        counts[r["ref_id"]][1 if r["flag"] & 4 else 0] += 1
    assert no_coordinate == unplaced
    for k, c in enumerate(contigs):
        for w, v in windows[k].items():
            assert c["linear"][w] == v, (k, w)
        if c["stats"] is None:
            assert counts[k] == [0, 0] and not c["bins"]
        else:
            assert list(c["stats"][2:]) == counts[k]
            assert c["stats"][0] == min(ch[0][0] for ch in c["bins"].values()) and c["stats"][1] == max(ch[-1][1] for ch in c["bins"].values())


def bam_reg2bin(beg, end):
    end -= 1
    for shift, first in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return first + (beg >> shift)
    return 0


@pytest.mark.parametrize("level", [0, 1])
def test_bam_index_matches_the_oracle_and_the_bai_semantics(level):
    from test_bam import make_tiles
    params, contigs, tiles = make_tiles(n_tiles=3, n_clusters=1500)
    o = oracle_lib.load()
    L = params.read_length[0]
    stream, n, unaligned = o.bam_records(tiles, [L, L])
    cuts = bam.split_parts(stream, unaligned)
    assert len(cuts) == len(contigs) + 1 and sum(c[1] for c in cuts) == len(stream)
    header_bytes = 977
    for order in ("back", "front"):
        ordered = cuts if order == "back" else cuts[-1:] + cuts[:-1]
        parts = [(off, size, bam.bgzf_compress(stream[off:off + size], level=level, n_threads=2)) for off, size in ordered]
        got = bam.index(stream, parts, len(contigs), header_bytes)
        assert got == o.bam_index(stream, parts, len(contigs), header_bytes)
        check_index_semantics(got, stream, parts, len(contigs), header_bytes, L)
    # a contig in several parts (the reference's bins are narrower than a contig), a contig without records, nothing at all
    off, size = cuts[0]
    half = 0
    while half < size // 2:
        half += 4 + int.from_bytes(stream[off + half:off + half + 4], "little")
    split = [(off, half), (off + half, size - half)] + cuts[1:]
    parts = [(a, b, bam.bgzf_compress(stream[a:a + b], level=level)) for a, b in split]
    got = bam.index(stream, parts, len(contigs) + 2, header_bytes)
    assert got == o.bam_index(stream, parts, len(contigs) + 2, header_bytes)
    assert bam.parse_index(got)[0][-1] == dict(bins={}, stats=None, linear=[])
    assert bam.index(b"", [], 2, 100) == o.bam_index(b"", [], 2, 100) == b"BAI\1" + (2).to_bytes(4, "little") + bytes(16) + bytes(8)
    with pytest.raises(bam.BamError):
        bam.index(stream, [(cuts[0][0], cuts[0][1], b"not bgzf at all")], len(contigs), 0)
    with pytest.raises(bam.BamError):
        bam.index(stream, [(o_, s, bam.bgzf_compress(stream[o_:o_ + s])) for o_, s in (cuts[1], cuts[0])], len(contigs), 0)       # out of contig order


# ---- the whole program -----------------------------------------------------------------------------------------------------------------

def write_fasta(path, names, contigs, rng):
    """contigs with lower case stretches and IUPAC codes in the file; returns (per contig byte offset / size in the file, the contigs as
    reference::loadContig reads them: upper case, everything that is not ACGT an N)"""
    meta, loaded = [], []
    with open(path, "wb") as f:
        for name, c in zip(names, contigs):
            text = bytearray(c)
            for _ in range(5):
                at = int(rng.integers(0, len(text) - 300))
                text[at:at + 200] = bytes(text[at:at + 200]).lower()
            for at in rng.integers(0, len(text), 12):
                text[int(at)] = b"RYKMswbdhvn"[int(at) % 11]
            f.write(b">" + name + b" test contig\n")
            begin = f.tell()
            for at in range(0, len(text), 60):
                f.write(text[at:at + 60] + b"\n")
            meta.append((begin, f.tell() - begin))
            loaded.append(bytes(b if b in b"ACGT" else ord("N") for b in bytes(text).upper()))
    return meta, loaded


def plan_bins(lengths, bin_bases):
    """host/isaac_align.cpp: planBins -- ([(first, end) ReferencePosition values of every bin with positions], [cut positions inside contigs])"""
    bin_bases = int(min(1e15, bin_bases))
    bin_bases = max(2048, -(-bin_bases // 2048) * 2048)
    ranges, cuts, filled = [], [], 0
    for c, length in enumerate(lengths):
        if length > bin_bases + bin_bases // 2:
            pieces = -(-length // bin_bases)
            stretch = -(-(-(-length // pieces)) // 2048) * 2048
            for at in range(0, length, stretch):
                if at:
                    cuts.append(bam.reference_position(c, at))
                ranges.append((bam.reference_position(c, at), bam.reference_position(c, at + stretch) if at + stretch < length else bam.reference_position(c + 1, 0)))
            filled = 0
            continue
        if not filled or filled + length > bin_bases:
            ranges.append((bam.reference_position(c, 0), bam.reference_position(c + 1, 0))); filled = 0
        else:
            ranges[-1] = (ranges[-1][0], bam.reference_position(c + 1, 0))
        filled += max(length, 1)
    return ranges, cuts


def split_by_bins(record_bytes, unaligned_offset, bin_ranges):
    """[(offset, bytes)] of the records of every (bin, contig) that has some, then of the unaligned ones: the runs of BGZF blocks of the file"""
    data = memoryview(record_bytes)
    parts, at, b = [], 0, 0
    while at < unaligned_offset:
        ref_id, pos = int.from_bytes(data[at + 4:at + 8], "little"), int.from_bytes(data[at + 8:at + 12], "little")
        key = bam.reference_position(ref_id, pos)
        while not (bin_ranges[b][0] <= key < bin_ranges[b][1]):
            b += 1
        if not parts or parts[-1][2] != (b, ref_id):          # a run of BGZF blocks per bin and, inside a bin of several contigs, per contig
            parts.append([at, 0, (b, ref_id)])
        size = 4 + int.from_bytes(data[at:at + 4], "little")
        parts[-1][1] += size; at += size
    parts = [(p[0], p[1]) for p in parts]
    if unaligned_offset < len(data):
        parts.append((unaligned_offset, len(data) - unaligned_offset))
    return parts


SCENARIOS = {
    # the reference's defaults: y*n mask (101 -> 100 cycles), duplicates marked, gaps realigned, unaligned records at the back
    "defaults": dict(compressed=False, lengths=(100, 100), cli=[], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none"),
    "defaults-gz": dict(compressed=True, lengths=(100, 100), cli=[], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none"),
    # other masks and options: 75 + 80 cycles, nothing marked or realigned, unaligned records first, stored (level 0) BGZF, MAPQ 255 for unknown scores
    "options": dict(compressed=False, lengths=(75, 80),
                    cli=["--use-bases-mask", "y75n*,y80n*", "--keep-unaligned", "front", "--mark-duplicates", "0", "--realign-gaps", "no", "--bam-gzip-level", "0",
                         "--dodgy-alignment-score", "Unknown", "--bam-pu-format", "%F.%L"],
                    paired=True, mark=False, keep=True, realign=False, unaligned="front", dodgy=255, pu="%s.%d"),
    # two workers (contexts, threads) on the one device: loads, tiles and bins dealt between them, one file all the same
    "two-workers": dict(compressed=False, lengths=(100, 100), cli=["--devices", "0,0"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none"),
    # the bins' parts through host memory (what a run too large for the device's memory does), two workers
    "host-bins": dict(compressed=False, lengths=(100, 100), cli=["--devices", "0,0"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                      env={"ISAAC_ALIGN_HOST_BINS": "1"}),
    # ... and through files under --temp-directory (what a host does whose memory does not hold them: BinningFragmentStorage's bin files), one worker; the
    # second one by the option itself: -m 0 would mean no limit, so the limit is a gigabyte and nothing spills -- the switch is what the first one tests
    "spill-bins": dict(compressed=False, lengths=(100, 100), cli=["-m", "1"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                       env={"ISAAC_ALIGN_HOST_BINS": "1", "ISAAC_ALIGN_SPILL_BINS": "1"}, spills=True),
    # what two devices do, on one: the other contexts' contigs and table are copies (ISAAC_GPU_SHARE_BY_COPY) and each worker treats the other's blocks of bin parts as
    # another device's (ISAAC_ALIGN_STRANGERS): isaac_gpu_share_reference's copy branch and the fetch of foreign parts in the build stage
    "strangers": dict(compressed=False, lengths=(100, 100), cli=["--devices", "0,0"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                      env={"ISAAC_ALIGN_STRANGERS": "1", "ISAAC_GPU_SHARE_BY_COPY": "1"}),
    # the loads' BCL bytes wait in host memory for their selection (what a run does whose base calls do not fit the device beside the table)
    "host-loads": dict(compressed=False, lengths=(100, 100), cli=[], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                       env={"ISAAC_ALIGN_HOST_LOADS": "1"}, tiles_on_device=True),
    # bins of about 6 000 records: every contig is cut into several bins, each sorted, filtered and realigned by itself (the oracle with the same cuts)
    "cut-bins": dict(compressed=False, lengths=(100, 100), cli=["--bin-records", "6000", "--devices", "0,0"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0,
                     pu="%s:%d:none", bin_records=6000),
    # two devices for real (skipped where the box has one): a worker per device, the contigs and the table copied over the link, tiles and bins dealt between
    # them, foreign bin parts fetched in the build stage -- the same file as one device writes
    "two-devices": dict(compressed=False, lengths=(100, 100), cli=["--devices", "0,1"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                        needs_devices=2),
    # --default-adapters: a third of the pairs have inserts shorter than the reads, both reads run into the Nextera adapter and are clipped there
    # (FragmentSequencingAdapterClipper in the ungapped, gapped and rescue alignments; the oracle with the same list)
    "adapters": dict(compressed=False, lengths=(100, 100), cli=["--default-adapters", "Nextera"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                     adapters="Nextera"),
    # --realign-vigorously 1: a realigned fragment is tried again until nothing improves (GapRealigner.cpp:1241); the oracle's build stage likewise
    "vigorous": dict(compressed=False, lengths=(100, 100), cli=["--realign-vigorously", "1"], paired=True, mark=True, keep=True, realign=True, unaligned="back", dodgy=0, pu="%s:%d:none",
                     vigorous=True),
    # single-ended lanes, unaligned reads left out
    # ... on a reference made by bin/isaac-sort-reference from the FASTA file
    "single-ended": dict(compressed=True, lengths=(100,), cli=["--keep-unaligned", "discard"], paired=False, mark=True, keep=True, realign=True, unaligned="discard", dodgy=0, pu="%s:%d:none",
                         sort_reference_tool=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", sorted(SCENARIOS))
def test_gpu_isaac_align_end_to_end(tmp_path, scenario):
    import torch
    sc = SCENARIOS[scenario]
    if sc.get("needs_devices", 1) > torch.cuda.device_count():
        pytest.skip("needs %d GPUs" % sc["needs_devices"])
    compressed, lengths = sc["compressed"], sc["lengths"]
    n_reads, cluster_length = len(lengths), sum(lengths)
    o = oracle_lib.load()
    rng = np.random.default_rng(17)
    file_length, at_a_time = 101, 4000
    genome = synth.make_genome(300000, seed=61, n_contigs=3)
    names = [b"chrA", b"chrB", b"chrC"]
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    fasta = str(ref_dir / "genome.fa")
    meta, stored = write_fasta(fasta, names, [bytes(c.numpy()) for c in genome], rng)
    params = options.default_params(lengths[0], lengths[1] if n_reads > 1 else 0, dodgy_alignment_score=sc["dodgy"], keep_unaligned=int(sc["unaligned"] != "discard"))
    if sc.get("adapters"):
        options.set_adapters(params, sc["adapters"])
    a = gpu.Aligner(options.default_params(100, 100), 0, stored)
    a.build_index()
    xml = str(ref_dir / "sorted-reference.xml")
    if sc.get("sort_reference_tool"):
        # isaac-sort-reference: the contig table of printContigs, the same table as the library builds, in the files isaac-align reads
        karyotype = [0, 1, 2]
        r = subprocess.run([os.path.join(os.path.dirname(host()), "isaac-sort-reference"), "-g", fasta, "-o", str(ref_dir), "-j", "4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        contig_meta, masks, _ = sr.parse(open(xml).read())
        import hashlib
        position = 0
        for i, m in enumerate(contig_meta):
            sequence = bytes(b for b in open(fasta, "rb").read()[meta[i][0]:meta[i][0] + meta[i][1]].upper() if not chr(b).isspace())
            assert (m.name, m.index, m.karyotype_index, m.file.decode(), m.offset, m.size, m.total_bases, m.acgt_bases, m.genomic_position, m.bam_m5.decode()) == \
                   (names[i], i, i, fasta, meta[i][0], meta[i][1], len(stored[i]), sum(stored[i].count(b) for b in b"ACGT"), position, hashlib.md5(sequence).hexdigest())
            position += len(stored[i])
        files = sorted(f for f in os.listdir(ref_dir) if f.endswith(".dat"))
        assert len(files) == 64 and files[5] == "genome.fa-32mer-6bit-ABCD-05.dat" and [m.file.decode() for m in masks if m.seed_length == 32][5].endswith(files[5])
        assert np.concatenate([np.fromfile(ref_dir / f, abi.REFERENCE_KMER_DTYPE) for f in files]).tobytes() == a.get_index().tobytes()
    else:
        karyotype = [2, 0, 1]                                                # stored contig i is the karyotype[i]-th of the karyotype
        contig_meta, position = [], 0
        for i, c in enumerate(stored):
            m = sr.Contig()
            m.genomic_position, m.index, m.karyotype_index, m.name, m.file = position, i, karyotype[i], names[i], fasta.encode()
            m.offset, m.size, m.total_bases, m.acgt_bases = meta[i][0], meta[i][1], len(c), sum(c.count(b) for b in b"ACGT")
            if i == 1:
                m.bam_sq_as, m.bam_sq_ur, m.bam_m5 = b"testAssembly", b"http://example.org/chrB.fa", b"0123456789abcdef0123456789abcdef"
            position += len(c)
            contig_meta.append(m)
        a.save_sorted_reference(str(ref_dir), "genome.fa", contig_meta)
    ordered = [None] * 3
    for i, k in enumerate(karyotype):
        ordered[k] = i
    contigs = [stored[i] for i in ordered]                                   # karyotype order: the order of contig ids in records and of the header
    del a
    # two lanes (1 and 3) of one flowcell, read from a sample that carries an indel every ~400 bases (reads that cross one near their end
    # can borrow the gap from the reads that show it: work for the realigner); a tenth of the fragments of lane 1 sequenced twice
    sample = synth.make_sample_with_indels(contigs, rng)
    lanes = []
    for lane, n_pairs, seed in ((1, 14000, 62), (3, 9000, 63)):
        bcl = synth.make_read_pairs(sample, n_pairs, file_length, seed=seed, indel_read_fraction=0.01, n_rate=0.002)[0].numpy()
        if lane == 1:
            bcl[12600:14000] = bcl[rng.integers(0, 12000, 1400)]
        if sc.get("adapters"):
            from parity_util import add_adapters
            bcl, inserts = add_adapters(bcl, file_length, adapter=options.ADAPTER_PRESETS[sc["adapters"]][0][0], fraction=0.35, seed=seed + 7, insert_range=(45, 96))
            assert (inserts > 0).sum() > 2000
        lanes.append((lane, bcl))
    calls = tmp_path / "calls"
    calls.mkdir()
    texts = {}
    for lane, bcl in lanes:
        for read in range(n_reads):
            text = synth.bcl_to_fastq(bcl, read * file_length, file_length, name="M7:15:FCTEST:%d" % lane, plus_header=bool(read))
            texts[(lane, read)] = text
            path = calls / ("lane%d_read%d.fastq%s" % (lane, read + 1, ".gz" if compressed else ""))
            if compressed:
                half = text.rfind(b"\n@", 0, len(text) // 2) + 1
                path.write_bytes(gzip.compress(text[:half], 1) + gzip.compress(text[half:], 6))      # concatenated members
            else:
                path.write_bytes(text)
    out = tmp_path / "Aligned"
    args = ["-r", xml, "-b", str(calls), "--base-calls-format", "fastq-gz" if compressed else "fastq", "-o", str(out), "--clusters-at-a-time", str(at_a_time), "-j", "4",
            "--bam-header-tag", "@CO\tend to end", "--description", "cli test", "-t", str(tmp_path / "Temp")] + sc["cli"]
    r = run_host(*args, env=sc.get("env"))
    assert r.returncode == 0, r.stderr
    # nothing is left under --temp-directory (the bins' files of a run that spills are removed bin by bin as the file is written)
    assert not (tmp_path / "Temp").exists() or not list((tmp_path / "Temp").iterdir())
    timing = json.loads([l for l in r.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
    n_tiles = sum(-(-len(bcl) // at_a_time) for _, bcl in lanes)
    assert timing["tiles"] == n_tiles and timing["loads"] == n_tiles and timing["overflow_clusters"] == 0
    host_bins = "ISAAC_ALIGN_HOST_BINS" in sc.get("env", {})
    assert timing["tiles_kept_on_device"] == (0 if host_bins else n_tiles)                              # every tile's parts stayed on the device, or none did
    assert (timing["spilled_bytes"] > 0) == bool(sc.get("spills"))
    assert timing["loads_kept_on_device"] == (0 if "ISAAC_ALIGN_HOST_LOADS" in sc.get("env", {}) else n_tiles)
    # the bins the host made (host/isaac_align.cpp: planBins): contigs in karyotype order, grouped or cut by the reads the run was expected to have per base
    # (the host sizes its bins from an estimate of the cluster count -- file size over the length of the first record -- which it reports)
    ordered_lengths = [len(stored[i]) for i in sorted(range(3), key=lambda i: karyotype[i])]
    total_clusters = sum(len(b_) for _, b_ in lanes)
    assert 0.5 * total_clusters < timing["estimated_clusters"] < 8 * total_clusters
    bin_ranges, cuts = plan_bins(ordered_lengths, sc.get("bin_records", 4000000) / (timing["estimated_clusters"] * n_reads / sum(ordered_lengths)))
    assert timing["bin_cuts"] == len(cuts) and timing["bins"] == len(bin_ranges) + 1 and [tuple(r) for r in timing["bin_ranges"]] == bin_ranges
    assert (len(cuts) >= 6) if sc.get("bin_records") else (len(bin_ranges) == 1)
    # ---- the oracle on the same inputs
    b = gpu.Aligner(options.default_params(100, 100), 0, contigs)
    b.load_sorted_reference(xml)
    table = b.get_index()                                                  # as stored in the mask files: contig ids are stored indexes (the lookup translates them)
    del b
    position = table["position"].copy()
    located = (position >> np.uint64(1)) != 0                              # not a TooManyMatch entry
    stored_id = ((position[located] >> np.uint64(41)) - np.uint64(1)).astype(np.int64)
    position[located] = (position[located] & np.uint64((1 << 41) - 1)) | ((np.array(karyotype, np.uint64)[stored_id] + np.uint64(1)) << np.uint64(41))
    table["position"] = position
    ref = o.reference(contigs)
    ref.set_index(table)
    found, all_hits, index = [], np.zeros(3, np.uint8), 0
    for lane_index, (lane, bcl) in enumerate(lanes):
        o_lane = None
        for read in range(n_reads):
            rc, o_lane, n, _, _ = o.fastq_to_bcl(texts[(lane, read)], lengths[read], bcl=o_lane, cluster_stride=cluster_length, offset=sum(lengths[:read]), max_clusters=len(bcl))
            assert rc == 0 and n == len(bcl)
        number = 1
        for first in range(0, len(bcl), at_a_time):                          # loads of --clusters-at-a-time, each one tile
            tile_bcl = o_lane[first:first + at_a_time]
            om, hits = ref.find_matches(params, tile_bcl, len(tile_bcl), tile=index)
            all_hits |= hits
            found.append((lane_index, lane, number, index, tile_bcl, om))
            number += 1; index += 1
    host_tiles, tls_of_lane = [], {}
    for lane_index, lane, number, index, tile_bcl, om in found:
        tls = tls_of_lane.get(lane_index)
        if tls is None or not tls.stable:
            tls = tls_of_lane[lane_index] = ref.determine_tls(params, tile_bcl, om, all_hits, tile=index)
        orec, ocig, _ = ref.select(params, tile_bcl, om, tls, all_hits, tile=index, n_clusters_hint=len(tile_bcl))
        host_tiles.append((tile_bcl, orec, ocig, "FCTEST:%d:%d:" % (lane, number), str(lane_index), tls))
    want, want_n, want_unaligned = o.bam_records(host_tiles, list(lengths), forced_dodgy_alignment_score=sc["dodgy"] & 0xff, mark_duplicates=sc["mark"], keep_duplicates=sc["keep"],
                                                 realign_gaps=sc["realign"], reference=ref, bin_cuts=cuts, realign_vigorously=bool(sc.get("vigorous")))
    recs = bam.parse_records(want)
    print("oracle: %d records, %d unmapped, %d with gaps in the CIGAR, %d realigned, tiles %s" % (
        len(recs), sum(1 for x in recs if x["flag"] & 4), sum(1 for x in recs if any((int(w) & 15) in (1, 2) for w in x["cigar"])), sum(1 for x in recs if "OC" in x["tags"]),
        [(t[3], t[5].astuple(), int((t[1]["gap_count"] > 0).sum())) for t in host_tiles]))
    assert {x["tags"]["RG"] for x in recs} == {"0", "1"}
    if scenario.startswith("defaults"):
        assert sum(1 for x in recs if x["flag"] & 0x400) > 500 and sum(1 for x in recs if "OC" in x["tags"]) > 5
    if sc["unaligned"] == "discard":
        assert want_unaligned == len(want) and len(recs) < sum(len(b_) for _, b_ in lanes) * n_reads
    else:
        assert want_unaligned < len(want)
    sq = [(names[i].decode(), len(stored[i]), contig_meta[i].bam_sq_as.decode(), contig_meta[i].bam_sq_ur.decode() or fasta, contig_meta[i].bam_m5.decode()) for i in ordered]
    header = o.bam_header(" ".join([host()] + args), "isaac_aligner_amd-0.3", sq, description="cli test",
                          header_lines=["@CO\tend to end"] + ["@RG\tID:%d\tPL:ILLUMINA\tSM:default\tPU:%s" % (k, sc["pu"] % ("FCTEST", lane)) for k, (lane, _) in enumerate(lanes)])
    # ---- the files: the header, a BGZF run per contig and one for the unaligned records (first with --keep-unaligned front), the empty block
    cuts = split_by_bins(want, want_unaligned, bin_ranges)
    if sc["unaligned"] == "front" and want_unaligned < len(want):
        cuts = cuts[-1:] + cuts[:-1]
    expected = b"".join(want[off:off + size] for off, size in cuts)
    path = out / "Projects" / "default" / "default" / "sorted.bam"
    data = path.read_bytes()
    blocks = bgzf_blocks(data)
    raw = b"".join(x[2] for x in blocks)
    assert raw[:len(header)] == header
    assert raw[len(header):] == expected
    assert blocks[-1][1] == 28 and blocks[-1][2] == b""                      # bam::serializeBgzfFooter
    assert gzip.decompress(data) == raw
    starts, at = {}, 0
    for offset, size, block in blocks:
        starts.setdefault(at, offset); at += len(block)
    header_bgzf = starts[len(header)]
    parts, file_at = [], len(header)
    for off, size in cuts:
        parts.append((off, size, data[starts[file_at]:starts[file_at + size]]))
        file_at += size
    assert sum(len(p[2]) for p in parts) + header_bgzf + 28 == len(data)
    bai = (out / "Projects" / "default" / "default" / "sorted.bam.bai").read_bytes()
    assert bai == o.bam_index(want, parts, 3, header_bgzf)
    check_index_semantics(bai, want, parts, 3, header_bgzf, lengths[0])
