"""Pins the CPU restatement (oracle/) against the reference's own cppunit known-answer vectors (tests/golden/*.json,
generated from /root/reference by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle_lib import cigar_string

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def test_bsw_known_answers(oracle):
    g = load("bsw.json")
    bad = []
    for c in g["cases"]:
        cigar, _ = oracle.bsw_align(g["scores"], g["max_read_length"], c["query"], c["database"])
        if list(cigar) != c["cigar"]:
            bad.append((c["name"], c["genome"], cigar_string(cigar), cigar_string(c["cigar"])))
    assert not bad, bad[:10]


def test_bsw_avx2_rows_equal_the_lane_by_lane_rows(oracle):
    """the oracle's banded Smith-Waterman computes its rows sixteen lanes at a time (AVX2: what cpu_baseline times); the lane-by-lane restatement of the
    reference's statements is the form the known answers above were first pinned with: both on the reference's cases and on random ones, both presets,
    reads with substitutions, insertions, deletions and N"""
    g = load("bsw.json")
    cases = [(g["scores"], g["max_read_length"], c["query"].encode(), c["database"].encode()) for c in g["cases"]]
    rng = np.random.default_rng(5)
    for k in range(4000):
        scores = [[0, -3, 11, 4], [2, -1, 15, 3]][k % 2]
        L = int(rng.integers(20, 260))
        db = rng.integers(0, 4, L + 15 + 20)
        q = list(db[7:7 + L + 12])
        for _ in range(int(rng.integers(0, 6))):
            at = int(rng.integers(0, len(q) - 1))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                q[at] = (q[at] + 1 + int(rng.integers(0, 3))) % 4
            elif kind == 1:
                del q[at:at + 1 + int(rng.integers(0, 7))]
            else:
                q[at:at] = list(rng.integers(0, 4, 1 + int(rng.integers(0, 7))))
        q = (q + list(rng.integers(0, 4, L)))[:L]
        text = bytes(b"ACGT"[v] for v in q)
        if k % 7 == 0:
            text = text[:L // 2] + b"n" + text[L // 2 + 1:]
        data = bytes(b"ACGTN"[v if k % 11 or i != 9 else 4] for i, v in enumerate(db[:L + 15]))
        cases.append((scores, 300, text, data))
    for scores, max_len, q, d in cases:
        oracle.bsw_force_scalar(True)
        want = oracle.bsw_align(scores, max_len, q, d)
        oracle.bsw_force_scalar(False)
        got = oracle.bsw_align(scores, max_len, q, d)
        assert list(want[0]) == list(got[0]) and want[1] == got[1], (scores, q, d)


def test_bsw_overflow_rule(oracle):
    for m, mm, go, ge, length, throws in load("bsw.json")["overflow"]:
        assert oracle.bsw_check(m, mm, go, ge, length) == throws


def test_bsw_survey_smoke_vectors(oracle):
    # the unmodified reference object gives these on the strings of the (disabled) testCustom
    q = "CTAAGACCCCACACTCTGGGACACCAAGGTGGGAGGATCGCTGGAGCTCAGGAGTTTGAGACCAGCCTGGACAACATGGTGTGACCCTGTCTACAGAAAA"
    d = "AATGCCTCTGGCCTGGGCGTGGGAGTTCATGCTTGTAATCGCATATCGCTAGAGCCCAGGAGTTTGAGACCAGCCTGGACAACATGGTGAAAACCCTCGTTGCTACTAAAAATAC"
    cigar, off = oracle.bsw_align([2, -1, 15, 3], 300, q, d)
    assert (cigar_string(cigar), off) == ("100M", 8)
    cigar, off = oracle.bsw_align([0, -3, 11, 4], 300, q, d)
    assert (cigar_string(cigar), off) == ("27M3I70M", 11)


def test_seed_id(oracle):
    g = load("seed_id.json")
    shifts = {"reverse": 0, "seed": 1, "cluster": 9, "barcode": 40, "tile": 52}
    for v in g["valid"]:
        value = oracle.seed_id(*v)
        assert value == sum(x << shifts[k] for x, k in zip(v, g["order"]))
    for v in g["throws"]:
        with pytest.raises(RuntimeError):
            oracle.seed_id(*v)


def test_simple_indel_aligner(oracle):
    g = load("simple_indel.json")
    for i, c in enumerate(g["cases"]):
        out, cig = oracle.simple_indel_literal(c["read"], c["reference"], c["seeds"], c["left_clip0"], c["right_clip1"])
        f = out[0]
        e = c["expect"]
        got = cigar_string(cig[f["cigar_offset"]:f["cigar_offset"] + f["cigar_length"]])
        assert got == e["getCigarString"], (i, got, e)
        if "getMismatchCount" in e:
            assert f["mismatch_count"] == e["getMismatchCount"], (i, e)
        if "getEditDistance" in e:
            assert f["edit_distance"] == e["getEditDistance"], (i, e)
        if "getFStrandReferencePosition" in e:
            assert f["position"] == e["getFStrandReferencePosition"], (i, e)
        if "leftClipped" in e:
            assert f["low_clipped"] == e["leftClipped"] and f["high_clipped"] == e["rightClipped"], (i, e)


def test_fragment_builder2(oracle):
    g = load("fragment_builder2.json")
    for c in g["cases"]:
        f, cig, cyc = oracle.fragment_builder2_literal(c["read"], c["reference"], c["reverse"], c["position"], c["gapped"])
        e = c["expect"]
        assert cigar_string(cig) == e["getCigarString"], (c["name"], cigar_string(cig))
        assert f["mismatch_count"] == e["getMismatchCount"], c["name"]
        assert f["edit_distance"] == e["getEditDistance"], c["name"]
        assert f["observed_length"] == e["getObservedLength"], c["name"]
        if "getFStrandReferencePosition" in e:
            assert [f["contig_id"], f["position"]] == e["getFStrandReferencePosition"], c["name"]
        if "getStrandReferencePosition" in e:
            assert [f["contig_id"], f["position"]] == e["getStrandReferencePosition"], c["name"]
        if "getMismatchCyclesBegin" in e:
            assert cyc == e["getMismatchCyclesBegin"], c["name"]


def test_sequencing_adapter_known_answers(oracle):
    """lib/alignment/cppunit/testSequencingAdapter.cpp: the fifteen cases testEverything() runs -- mate-pair (bounded) and standard (unbounded, strand-bound)
    Nextera adapters found from the first mismatch on, the side to clip chosen by length / matches / the 40 % rule, adapters that start before the read or
    end behind it -- through FragmentSequencingAdapterClipper::checkInitStrand + UngappedAligner::alignUngapped of the oracle"""
    g = load("sequencing_adapter.json")
    assert len(g["cases"]) == 15
    for c in g["cases"]:
        f, cig = oracle.sequencing_adapter_literal(c["read"], c["reference"], c["reverse"], g["adapter_lists"][c["adapters"]])
        e = c["expect"]
        assert cigar_string(cig) == e["getCigarString"], (c["name"], cigar_string(cig))
        for key, field in (("getMismatchCount", "mismatch_count"), ("getEditDistance", "edit_distance"), ("getObservedLength", "observed_length")):
            if key in e:
                assert f[field] == e[key], (c["name"], key, f[field])
        for key in ("getFStrandReferencePosition", "getStrandReferencePosition"):
            if key in e:
                assert [f["contig_id"], f["position"]] == e[key], (c["name"], key)


def test_threaded_seed_lookup_equals_single_thread(oracle):
    """oracle_find_matches_mt (the CPU-baseline form: cluster ranges on host threads) returns the single-thread result"""
    from parity_util import make_inputs
    from isaac_aligner_amd import options
    contigs, bcl, _ = make_inputs(genome_bases=200000, n_pairs=3001, seed=9)
    ref = oracle.reference(contigs)
    ref.build_index()
    p = options.default_params(150, 150)
    m1, h1 = ref.find_matches(p, bcl, len(bcl))
    m7, h7 = ref.find_matches(p, bcl, len(bcl), n_threads=7)
    assert len(m1) == len(m7) and (m1 == m7).all() and (h1 == h7).all()


def test_template_length_statistics_known_answers(oracle):
    """testTemplateLengthStatistics.cpp:42-367: alignment models and classes, mate orientation and mate position windows for the
    eight model pairs, and the statistics of the literal addTemplates() sequence (min 14, median 5001, max 9987, sd 3414/3413)"""
    import ctypes as C
    g = json.load(open(os.path.join(GOLDEN, "template_length_statistics.json")))
    lib = oracle.lib
    model_names = {0: "FF+", 1: "FR+", 2: "RF+", 3: "RR+", 4: "FF-", 5: "FR-", 6: "RF-", 7: "RR-"}       # TemplateLengthStatistics.cpp alignmentModelName
    class_names = {0: "F+", 1: "R+", 2: "R-", 3: "F-"}
    for c in g["alignment_models"]:
        m = lib.oracle_tls_alignment_model(C.c_int64(c["f1"][0]), c["f1"][1], C.c_int64(c["f2"][0]), c["f2"][1])
        assert model_names[m] == c["name"], c
    for c in g["alignment_classes"]:
        assert class_names[lib.oracle_tls_alignment_class(g["models"][c["model"]])] == c["name"], c
    out = (C.c_int64 * 3)()
    for c in g["mates"]:
        lens = c.get("read_lengths", [67, 83])
        lib.oracle_tls_mate(*[C.c_uint32(v) for v in c["stats"]], g["models"][c["models"][0]], g["models"][c["models"][1]], c["drift"],
                            C.c_uint32(c["read_index"]), int(c["reverse"]), C.c_int64(c.get("position", 500)), C.c_uint32(lens[0]), C.c_uint32(lens[1]), out)
        got = {"orientation": out[0], "min_position": out[1], "max_position": out[2]}[c["what"]]
        assert got == c["expected"], (c, got)
    seq = (C.c_uint32 * 10)()
    exp = g["add_templates"]["after_10000"]
    lib.oracle_tls_add_templates_sequence(-1, seq)
    assert [seq[0], seq[1], seq[2], seq[3], seq[4]] == [exp["Min"], exp["Median"], exp["Max"], exp["LowStdDev"], exp["HighStdDev"]]
    assert (seq[5], seq[6]) == (exp["Min"], exp["Max"])                # testNoMateDriftRange
    assert seq[7] == 0 and seq[8] == 1 and seq[9] == 1                 # every addTemplate false until the last one
    drift = g["add_templates"]["mate_drift_range"]
    lib.oracle_tls_add_templates_sequence(drift, seq)
    assert seq[1] == exp["Median"] and (seq[5], seq[6]) == (exp["Median"] - drift, exp["Median"] + drift)     # testMateDriftRange


def test_end_clippers_known_answers(oracle):
    """testSemialignedClipper.cpp:189-251 (four literal alignments) and testOverlappingEndsClipper.cpp:109-157 (two literal pairs)"""
    import ctypes as C
    from isaac_aligner_amd import abi
    g = json.load(open(os.path.join(GOLDEN, "clippers.json")))
    lib = oracle.lib
    cig, n, pos = (C.c_uint32 * 16)(), C.c_uint64(), C.c_int64()
    for c in g["semialigned"]:
        assert 0 == lib.oracle_semialigned_clip_literal(c["read"].encode(), c["reference"].encode(), int(c["reverse"]), cig, C.c_uint64(16), C.byref(n), C.byref(pos))
        assert abi.cigar_string(list(cig[:n.value])) == c["cigar"] and pos.value == c["position"], (c, abi.cigar_string(list(cig[:n.value])), pos.value)
    cig2, n2, pos2 = (C.c_uint32 * 16)(), (C.c_uint32 * 2)(), (C.c_int64 * 2)()
    for c in g["overlapping"]:
        assert 0 == lib.oracle_overlapping_clip_literal(c["read1"].encode(), c["quality1"].encode(), int(c["reverse1"]), c["read2"].encode(), c["quality2"].encode(), int(c["reverse2"]),
                                                        c["reference"].encode(), cig2, n2, pos2)
        for i in (0, 1):
            assert abi.cigar_string(list(cig2[8 * i:8 * i + n2[i]])) == c["cigar"][i] and pos2[i] == c["position"][i], (c, i, abi.cigar_string(list(cig2[8 * i:8 * i + n2[i]])), pos2[i])


def test_template_builder_known_answers(oracle):
    """testTemplateBuilder.cpp:149-373: the candidate lists of every test through TemplateBuilder::buildTemplate; every asserted
    value must come out, notably the alignment scores (1136 / 534 / 569, 1119 / 517, 1084, 2 / 2 / 3): the MAPQ arithmetic
    (rest-of-genome correction, exp / log10, shadow rescue probabilities) against the reference's own numbers"""
    import ctypes as C
    g = json.load(open(os.path.join(GOLDEN, "template_builder.json")))

    class Frag(C.Structure):
        _fields_ = [("contig_id", C.c_uint32), ("position", C.c_int64), ("observed_length", C.c_uint32), ("read_index", C.c_uint32), ("reverse", C.c_uint32),
                    ("cigar_offset", C.c_uint32), ("cigar_length", C.c_uint32), ("mismatch_count", C.c_uint32), ("log_probability", C.c_double),
                    ("unique_seed_count", C.c_uint32), ("alignment_score", C.c_uint32), ("no_match", C.c_uint32)]

    def pack(frags):
        a = (Frag * max(1, len(frags)))()
        for i, f in enumerate(frags):
            for k, v in f.items():
                setattr(a[i], k, int(v) if isinstance(v, bool) else v)
        return a

    lib = oracle.lib
    for fixture in g["fixtures"]:
        contigs = (C.c_char_p * len(fixture))(*[c.encode() for c in fixture])
        for case in g["cases"]:
            out, score = (Frag * 2)(), C.c_uint32()
            rc = lib.oracle_template_builder_literal(contigs, C.c_uint32(len(fixture)), C.c_uint32(g["bcl"]["contig"]), g["bcl"]["offset0"], g["bcl"]["offset1"],
                                                     pack(case["fragments0"]), C.c_uint32(len(case["fragments0"])), pack(case["fragments1"]), C.c_uint32(len(case["fragments1"])),
                                                     C.c_uint32(1), C.byref(score), out)
            assert rc == 0
            exp = case["expected"]
            assert score.value == exp["template_score"], (case["name"], score.value, exp["template_score"])
            for i in (0, 1):
                for k, v in exp["fragments"][i].items():
                    got = getattr(out[i], k)
                    assert got == (int(v) if isinstance(v, bool) else v), (case["name"], i, k, got, v)


def test_shadow_aligner_known_answers(oracle):
    """testShadowAligner.cpp:56-252: an orphan rescues its mate, the mate rescues the orphan back, in both orientations and on a
    short and a long contig: position, strand, observed length, CIGAR and log probability as asserted there"""
    import ctypes as C
    g = json.load(open(os.path.join(GOLDEN, "shadow_aligner.json")))
    assert tuple(g["read_lengths"]) == (81, 92)

    class Frag(C.Structure):
        _fields_ = [("contig_id", C.c_uint32), ("position", C.c_int64), ("observed_length", C.c_uint32), ("read_index", C.c_uint32), ("reverse", C.c_uint32),
                    ("cigar_offset", C.c_uint32), ("cigar_length", C.c_uint32), ("mismatch_count", C.c_uint32), ("log_probability", C.c_double),
                    ("unique_seed_count", C.c_uint32), ("alignment_score", C.c_uint32), ("no_match", C.c_uint32)]
    lib = oracle.lib
    for fixture in g["fixtures"]:
        contigs = (C.c_char_p * len(fixture))(*[c.encode() for c in fixture])
        for blk in g["blocks"]:
            out, words, ok = (Frag * 2)(), (C.c_uint32 * 2)(), (C.c_uint32 * 2)()
            t, b = blk["tls"], blk["bcl"]
            rc = lib.oracle_shadow_aligner_literal(contigs, C.c_uint32(len(fixture)), C.c_uint32(b["contig"]), b["offset0"], b["offset1"], int(b["reverse0"]), int(b["reverse1"]),
                                                   C.c_uint32(t["min"]), C.c_uint32(t["max"]), C.c_uint32(t["median"]), C.c_uint32(t["low_std_dev"]), C.c_uint32(t["high_std_dev"]),
                                                   t["model0"], t["model1"], int(blk["orphan_reverse"]), out, words, ok)
            assert rc == 0 and list(ok) == [1, 1], (blk, list(ok))
            for k, e in enumerate(blk["expected"]):
                f = out[k]
                assert f.contig_id == b["contig"] and f.read_index == (1 - k)
                # cigarOffset is 0 in the reference's run because the mate happens to be the first candidate of its window there; on
                # other draws of the rand() contigs chance 7-mer hits precede it, so the fragment's own CIGAR word is compared instead
                assert (f.position, bool(f.reverse), f.observed_length, f.mismatch_count, f.cigar_length, words[k]) == (
                    e["position"], e["reverse"], e["observed_length"], e["mismatch_count"], e["cigar_length"], e["first_cigar_word"]), (blk, k)
                assert abs(f.log_probability - e["log_probability"]) <= e["log_probability_tolerance"]


def test_fragment_builder_known_answers(oracle):
    """testFragmentBuilder.cpp:33-598: hand-made seed match lists -> FragmentBuilder::build candidates (single seed, seed offsets,
    several seeds of one alignment, repeats on two contigs, mismatches and their log probabilities, alignments hanging over
    either end of a short contig and the soft clips they get)"""
    from parity_util import check_fragment_builder_case, fragment_builder_inputs, fragment_builder_params
    g = load("fragment_builder.json")
    for k, fixture in enumerate(g["fixtures"]):
        ref = oracle.reference([c.encode() for c in fixture["contigs"]])
        for case in g["cases"]:
            if case["name"] == "testMismatches" and k not in g["mismatch_fixtures"]:
                continue
            p = fragment_builder_params(g, case["repeat_threshold"])
            bcl, matches, tile = fragment_builder_inputs(case, fixture, oracle.seed_id)
            cands, cigars = ref.build_fragments(p, bcl, matches, tile=tile, with_gaps=case["with_gaps"], trim=False)
            check_fragment_builder_case(case, cands, cigars)
    # testEmptyMatchList (:84-104): nothing in, nothing out
    cands, cigars = ref.build_fragments(p, bcl, matches[:0], tile=tile, with_gaps=True, trim=False)
    assert len(cands) == 0 and len(cigars) == 0


def test_cluster_info_kmer_generator_permutate_and_neighbors_finder(oracle):
    """The reference's small unit tests behind the seed lookup (ClusterInfo: which reads still get seeds), the mate rescue (7-mer
    streams of KmerGenerator) and the index builder's neighbour annotation (Permutate, the 70-permutation list, findNeighbors):
    testMatchFinderClusterInfo.cpp, testKmerGenerator.cpp, testPermutate.cpp, testNeighborsFinder.cpp"""
    import ctypes as C
    import numpy as np
    g = load("oligo.json")
    lib = oracle.lib
    lib.oracle_max_kmer.restype = C.c_uint64
    # ---- ClusterInfo: replay the statements of the test in order
    objects = {}

    def state(var):
        ops = objects[var]
        st = (C.c_uint32 * 6)()
        oracle.check(lib.oracle_cluster_info(-1, (C.c_uint32 * len(ops))(*[o for o, _ in ops]), (C.c_uint32 * len(ops))(*[a for _, a in ops]), C.c_uint32(len(ops)), st))
        return {"getBarcodeIndex": st[0], "isBarcodeSet": bool(st[1]), "isReadComplete": [bool(st[2]), bool(st[3])], "bytes": (st[4], st[5])}
    for step in g["cluster_info"]:
        if step["op"] == "new": objects[step["var"]] = []
        elif step["op"] == "markReadComplete": objects[step["var"]].append((0, step["arg"]))
        elif step["op"] == "setBarcodeIndex": objects[step["var"]].append((1, step["arg"]))
        else:
            got = state(step["var"])[step["what"]]
            if step["what"] == "isReadComplete": got = got[step["arg"]]
            assert got == step["expected"], step
    assert state("none")["bytes"] == (0xfe, 0xfe) and state("all")["bytes"] == (0xff, 0xff)       # the layout drawn at TileClusterInfo.hh:55-64
    # ---- KmerGenerator
    for s in g["kmer_generator"]["streams"]:
        kmers, positions, n = np.zeros(64, np.uint32), np.zeros(64, np.int64), C.c_uint64()
        oracle.check(lib.oracle_kmer_generator(s["sequence"].encode(), C.c_uint64(len(s["sequence"])), C.c_uint32(s["k"]), kmers.ctypes.data_as(C.c_void_p),
                                               positions.ctypes.data_as(C.c_void_p), C.c_uint64(64), C.byref(n)))
        assert list(kmers[:n.value]) == s["kmers"] and list(positions[:n.value]) == s["positions"], s
    for k, v in g["kmer_generator"]["max_kmer"]:
        assert lib.oracle_max_kmer(C.c_uint32(k)) == v
    for c in g["kmer_generator"]["generate_kmer"]:
        kmer = C.c_uint32()
        ok = lib.oracle_generate_kmer(C.c_uint32(c["k"]), c["sequence"].encode(), C.c_uint64(len(c["sequence"])), C.byref(kmer))
        assert bool(ok) == c["ok"] and (not c["ok"] or kmer.value == c["kmer"]), c
    # ---- Permutate
    hi_lo = lambda v: (C.c_uint64(v >> 64), C.c_uint64(v & (2 ** 64 - 1)))
    for b in g["permutate"]["blocks"]:
        for c in b["checks"]:
            n = len(c["from"])
            hi, lo = C.c_uint64(), C.c_uint64()
            oracle.check(lib.oracle_permutate(C.c_uint32(b["block_length"]), (C.c_uint32 * n)(*c["from"]), (C.c_uint32 * n)(*c["to"]), C.c_uint32(n), C.c_uint32(32), int(c["reorder"]),
                                              *hi_lo(int(c["kmer"], 16)), C.byref(hi), C.byref(lo)))
            assert lo.value == int(c["expected"], 16), (b["name"], c, hex(lo.value))
    for c in g["permutate"]["lists"]:
        size, hi, lo, ok = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        oracle.check(lib.oracle_permutate_list_walk(C.c_uint32(c["kmer_bases"]), C.c_uint32(c["error_count"]), *hi_lo(int(c["original"], 16)), C.byref(size), C.byref(hi), C.byref(lo), C.byref(ok)))
        assert size.value == c["size"] and ok.value == 1 and ((hi.value << 64) | lo.value) == int(c["expected"], 16), c
    # ---- NeighborsFinder::findNeighbors
    nf = g["neighbors_finder"]
    for run in nf["runs"]:
        values = [int(k, 16) for k in run["kmers"]]
        hi = np.array([v >> 64 for v in values], np.uint64); lo = np.array([v & (2 ** 64 - 1) for v in values], np.uint64)
        for jobs in (nf["jobs"], 1, 16):
            flags = np.zeros(len(values), np.uint8)
            oracle.check(lib.oracle_find_neighbors(C.c_uint32(run["kmer_bases"]), hi.ctypes.data_as(C.c_void_p), lo.ctypes.data_as(C.c_void_p), C.c_uint64(len(values)), C.c_uint32(jobs),
                                                   flags.ctypes.data_as(C.c_void_p)))
            assert [bool(f) for f in flags] == nf["expected"], (run["kmer_bases"], jobs, list(flags))


def test_duplicate_filter_reproduces_the_reference_test(oracle):
    """lib/build/cppunit/testDuplicateFiltering.cpp: every filterInput run of the suite (FDuplicateFilter / RSDuplicateFilter, keepDuplicates false)
    leaves exactly the entries the test expects"""
    g = json.load(open(os.path.join(GOLDEN, "duplicate_filtering.json")))
    assert len(g["cases"]) == 13
    for case in g["cases"]:
        entries = [g["entries"][name] for name in case["input"]]
        kinds = {e["kind"] for e in entries}
        assert len(kinds) == 1
        primary = [e["anchor"] if e["kind"] == "rs" else e["f_strand_pos"] for e in entries]
        dup = oracle.filter_duplicates(primary, [e["mate_anchor"] for e in entries], [e["mate_info"] for e in entries], [e["rank"] for e in entries], [e["cluster_id"] for e in entries])
        kept = sorted(name for name, d in zip(case["input"], dup) if not d)
        assert kept == sorted(case["expected_unique"]), (case["test"], kept, case["expected_unique"])


def test_gap_realigner_reproduces_the_reference_test(oracle):
    """lib/build/cppunit/testGapRealigner.cpp (testFull, testMore): the fragment, reference and gaps the fixture builds from the test's strings go
    through the oracle's GapRealigner::realign; original CIGAR / position / edit distance are the fixture's, realigned ones and the overlap masks
    of OverlappingGapsFilter are what the test asserts"""
    g = json.load(open(os.path.join(GOLDEN, "gap_realigner.json")))
    assert len(g["cases"]) == 62
    for k, case in enumerate(g["cases"]):
        e = case["expected"]
        assert cigar_string(case["cigar"]) == e["originalCigar_"], k
        if "originalPos_" in e:
            assert case["f_strand_position"] == e["originalPos_"], k
        if "originalEditDistance_" in e:
            assert case["edit_distance"] == e["originalEditDistance_"], k
        got = oracle.realign_case(case, g["realigner"])
        assert got["position"] == e["realignedPos_"], (k, got, e)
        if "realignedCigar_" in e:
            assert got["cigar"] == e["realignedCigar_"], (k, got, e)
        assert got["edit_distance"] == e["realignedEditDistance_"], (k, got, e)
        if "overlappingGapsFilter_.overlapsCount()" in e:
            assert len(got["overlaps"]) == e["overlappingGapsFilter_.overlapsCount()"], (k, got, e)
        for n in (0, 1):
            if "overlappingGapsFilter_.overlap(%d)" % n in e:
                assert got["overlaps"][n] == e["overlappingGapsFilter_.overlap(%d)" % n], (k, got, e)
