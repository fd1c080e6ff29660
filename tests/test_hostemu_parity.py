"""The product's thread-serial device logic (isaac_aligner_amd/csrc/*.h compiled for the CPU by tests/hostemu) against the
oracle, on seeded synthetic inputs.  This validates the host logic and the kernels' per-cluster functions without a GPU;
the same comparisons run on the real kernels in test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

import hostemu_lib
import oracle_lib
from isaac_aligner_amd import abi, options
from parity_util import compare_candidates, compare_records, make_inputs


@pytest.fixture(scope="module")
def emulib():
    return hostemu_lib.load()


def test_exact_sort_is_std_sort(emulib):
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 15, 16, 17, 33, 100, 257, 1000, 5000]:
        for _ in range(10):
            keys = rng.integers(0, max(2, n // 4 + 1), n).astype(np.uint32)
            a, b = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
            emulib.emu_exact_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(a))
            emulib.emu_std_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(b))
            assert (a == b).all(), n
    # adversarial: organ-pipe / sorted / constant inputs drive introsort into its heap-sort fallback
    for n in [64, 300, 2000]:
        for keys in (np.arange(n), np.arange(n)[::-1], np.zeros(n), np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]])):
            keys = np.ascontiguousarray(keys, np.uint32)
            a, b = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
            emulib.emu_exact_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(a))
            emulib.emu_std_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(b))
            assert (a == b).all(), n


def test_default_params_match_oracle(oracle):
    for lens in [(100, 100), (150, 150), (250, 250), (150, 0), (36, 36), (75, 101)]:
        n_reads = 2 if lens[1] else 1
        assert bytes(oracle.default_params(n_reads, lens[0], lens[1])) == bytes(options.default_params(*lens)), lens


@pytest.mark.parametrize("cfg", [
    dict(read_length=150, n_pairs=1500, seed=1),
    dict(read_length=100, n_pairs=1200, seed=5),
    dict(read_length=250, n_pairs=600, seed=9, indel_read_fraction=0.2, indel_max=10),
    dict(read_length=150, read_length2=100, n_pairs=800, seed=13, subst_rate=0.02, n_rate=0.01),
])
def test_fragments_tls_and_records(oracle, emulib, cfg):
    contigs, bcl, _ = make_inputs(**cfg)
    n = len(bcl)
    p = options.default_params(cfg["read_length"], cfg.get("read_length2") or cfg["read_length"])
    ref = oracle.reference(contigs)
    ref.build_index()
    matches, hits = ref.find_matches(p, bcl, n)
    emu = hostemu_lib.Emu(emulib, p, contigs, hits)
    emu.set_matches(matches, n)
    for with_gaps, trim in ((True, True), (False, False)):
        oc, ocig = ref.build_fragments(p, bcl, matches, hits, with_gaps=with_gaps, trim=trim)
        ec, ecig = emu.build_fragments(bcl, n, with_gaps=with_gaps, trim=trim)
        assert not compare_candidates(oc, ocig, ec, ecig)
    otls = ref.determine_tls(p, bcl, matches, hits)
    etls = emu.determine_tls(bcl, n)
    assert otls.astuple() == etls.astuple()
    orec, ocig, _ = ref.select(p, bcl, matches, otls, hits, n_clusters_hint=n)
    erec, ecig = emu.select(bcl, n, etls)
    assert not (erec["reserved"] & 5).any()
    assert not compare_records(orec, ocig, erec, ecig)
    assert emu.counters()["mapq_near_integer"] == 0


@pytest.mark.parametrize("adapters", ["Standard", "Nextera", "NexteraMp"])
def test_sequencing_adapters_through_the_device_code(oracle, emulib, adapters):
    """--default-adapters: a third of the pairs have inserts of 60-145 bases at 2 x 150, so both reads run into the adapter.  The device headers (adapter ranges per
    read and strand from the first candidate, the clip ahead of the ungapped scan, of the gapped window and of the rescue scans) against the oracle's
    FragmentSequencingAdapterClipper: candidates with and without gaps, template statistics, every record.  The clips must actually happen."""
    from parity_util import add_adapters
    text = dict(Standard="AGATCGGAAGAGC", Nextera="CTGTCTCTTATACACATCT", NexteraMp="CTGTCTCTTATACACATCT")[adapters]
    contigs, bcl, _ = make_inputs(read_length=150, n_pairs=1500, seed=21, indel_read_fraction=0.1)
    bcl, inserts = add_adapters(bcl, 150, adapter=text, adapter2="AGATGTGTATAAGAGACAG" if adapters == "NexteraMp" else None, fraction=0.35, seed=22)
    n = len(bcl)
    p = options.set_adapters(options.default_params(150, 150), adapters)
    plain = options.default_params(150, 150)
    ref = oracle.reference(contigs)
    ref.build_index()
    matches, hits = ref.find_matches(p, bcl, n)
    emu = hostemu_lib.Emu(emulib, p, contigs, hits)
    emu.set_matches(matches, n)
    for with_gaps, trim in ((True, True), (False, False)):
        oc, ocig = ref.build_fragments(p, bcl, matches, hits, with_gaps=with_gaps, trim=trim)
        ec, ecig = emu.build_fragments(bcl, n, with_gaps=with_gaps, trim=trim)
        assert not compare_candidates(oc, ocig, ec, ecig)
    otls = ref.determine_tls(p, bcl, matches, hits)
    etls = emu.determine_tls(bcl, n)
    assert otls.astuple() == etls.astuple()
    orec, ocig, _ = ref.select(p, bcl, matches, otls, hits, n_clusters_hint=n)
    erec, ecig = emu.select(bcl, n, etls)
    assert not (erec["reserved"] & 5).any()
    assert not compare_records(orec, ocig, erec, ecig)
    # the adapters are found: the records differ from a run without them, in the short-insert pairs and (nearly) nowhere else, and most of those are clipped
    # where the insert ends
    prec, pcig, _ = ref.select(plain, bcl, matches, ref.determine_tls(plain, bcl, matches, hits), hits, n_clusters_hint=n)
    differs = ((orec["f_strand_position"] != prec["f_strand_position"]) | (orec["observed_length"] != prec["observed_length"]) | (orec["cigar_length"] != prec["cigar_length"])).reshape(-1, 2).any(1)
    short = inserts > 0
    assert differs[short].mean() > 0.25 and differs[~short].mean() < 0.02, (differs[short].mean(), differs[~short].mean())
    aligned = (orec["flags"] & 4 == 0).reshape(-1, 2)[:, 0] & short
    at_insert = orec["observed_length"].reshape(-1, 2)[:, 0] == inserts
    assert (at_insert & aligned).sum() > 0.5 * aligned.sum(), ((at_insert & aligned).sum(), aligned.sum())


def test_template_code_reproduces_reference_known_answers(emulib):
    """the product's template code (device headers, thread-serial form) on the candidate lists of testTemplateBuilder.cpp: the
    reference's own alignment scores (1136 / 534 / 569, 1119 / 517, 1084, 2 / 2 / 3) and placements must come out"""
    import ctypes as C
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "template_builder.json")))

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

    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    # TemplateBuilder(flowcells, 10, 4, false, 8, false, ELAND scores, 20000, DODGY_ALIGNMENT_SCORE_UNALIGNED), no clipping in buildTemplate
    p = options.default_params(100, 100, gap_scoring="eland", gapped_mismatches_max=8, semialigned_gap_limit=20000, dodgy_alignment_score=-1,
                               clip_semialigned=0, clip_overlapping=0)
    tls = abi.Tls()
    tls.min, tls.max, tls.median, tls.low_std_dev, tls.high_std_dev, tls.stable, tls.mate_min, tls.mate_max = 150, 250, 190, 20, 30, 1, 150, 250
    tls.best_model[0], tls.best_model[1] = 1, 6       # FR+, RF-
    for fixture in g["fixtures"]:
        emu = hostemu_lib.Emu(emulib, p, [c.encode() for c in fixture])
        forward = fixture[g["bcl"]["contig"]]
        reverse = "".join(comp[b] for b in reversed(forward))
        bases = forward[g["bcl"]["offset0"]:g["bcl"]["offset0"] + 100] + reverse[g["bcl"]["offset1"]:g["bcl"]["offset1"] + 100]
        bcl = np.array([(40 << 2) | "ACGT".index(b) for b in bases], np.uint8)
        for case in g["cases"]:
            rec = np.zeros(2, abi.FRAGMENT_DTYPE)
            cig = np.zeros(2 * abi.MAX_CIGAR_OPS, np.uint32)
            rc = emulib.emu_select_literal(emu.h, hostemu_lib.ptr(bcl), pack(case["fragments0"]), C.c_uint32(len(case["fragments0"])), pack(case["fragments1"]),
                                           C.c_uint32(len(case["fragments1"])), C.byref(tls), hostemu_lib.ptr(rec), hostemu_lib.ptr(cig))
            assert rc == 0
            exp = case["expected"]
            if case["fragments0"] or case["fragments1"]:      # without candidates the cluster never reaches buildTemplate in the product path
                assert rec["template_alignment_score"][0] == (exp["template_score"] & 0xffff), (case["name"], rec["template_alignment_score"], exp["template_score"])
            for i in (0, 1):
                e = exp["fragments"][i]
                if "alignment_score" in e:
                    assert rec["alignment_score"][i] == (e["alignment_score"] & 0xffff), (case["name"], i, rec["alignment_score"][i], e["alignment_score"])
                if e.get("no_match"):
                    assert rec["flags"][i] & 2                      # unaligned
                    continue
                if "position" in e:
                    assert abi.refpos_position(rec["f_strand_position"][i:i + 1])[0] == e["position"], (case["name"], i)
                if "contig_id" in e:
                    assert abi.refpos_contig(rec["f_strand_position"][i:i + 1])[0] == e["contig_id"], (case["name"], i)
                if "observed_length" in e:
                    assert rec["observed_length"][i] == e["observed_length"], (case["name"], i)


def test_fragment_code_reproduces_reference_known_answers(oracle, emulib):
    """lib/alignment/cppunit/testFragmentBuilder.cpp:33-598 through the device headers (candidate building, ungapped alignment,
    contig-end soft clips, consolidation): the values the reference's test asserts, not only agreement with the oracle"""
    import json
    import os
    from parity_util import check_fragment_builder_case, fragment_builder_inputs, fragment_builder_params
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fragment_builder.json")))
    for k in g["mismatch_fixtures"][:2]:              # two draws of the contigs on which every test of the suite applies
        fixture = g["fixtures"][k]
        contigs = [c.encode() for c in fixture["contigs"]]
        for case in g["cases"]:
            p = fragment_builder_params(g, case["repeat_threshold"], device_limits=True)
            bcl, matches, _ = fragment_builder_inputs(case, fixture, oracle.seed_id)
            emu = hostemu_lib.Emu(emulib, p, contigs)
            emu.set_matches(matches, len(bcl))
            cands, cigars = emu.build_fragments(bcl, len(bcl), with_gaps=case["with_gaps"], trim=False)
            check_fragment_builder_case(case, cands, cigars)


@pytest.mark.parametrize("sums_cap,radix_min", [(1024, -1), (24, -1), (1024, 0), (1024, 20)])
def test_repeat_rich_reference_through_the_precomputed_sums(oracle, emulib, sums_cap, radix_min):
    """a small human-like reference (Alu / L1-like families, satellites, segmental duplications): clusters with dozens of seeded
    candidates and hundreds of rescued shadows go through the probability-sum stage (sums.h); with a small key capacity most of
    them take the wave-per-cluster route instead, and the records must not depend on the route"""
    from isaac_aligner_amd import synth
    g = synth.make_human_like_genome(1_500_000, seed=11)
    contigs = [bytes(c.numpy()) for c in g.contigs]
    bcl = synth.make_read_pairs(g, 1500, 150, seed=12, avoid_gaps=True)[0].numpy()
    n = len(bcl)
    p = options.default_params(150, 150)
    ref = oracle.reference(contigs)
    ref.build_index()
    matches, hits = ref.find_matches(p, bcl, n)
    emu = hostemu_lib.Emu(emulib, p, contigs, hits)
    emu.set_matches(matches, n)
    emulib.emu_set_sums_capacity(emu.h, C.c_uint32(sums_cap))
    emulib.emu_set_sums_radix(emu.h, C.c_int(radix_min))        # the radix ordering of the HBM tier (device: lists beyond 3584 entries)
    otls = ref.determine_tls(p, bcl, matches, hits)
    etls = emu.determine_tls(bcl, n)
    assert otls.astuple() == etls.astuple()
    orec, ocig, _ = ref.select(p, bcl, matches, otls, hits, n_clusters_hint=n)
    erec, ecig = emu.select(bcl, n, etls)
    assert not (erec["reserved"] & 5).any()
    assert not compare_records(orec, ocig, erec, ecig)
    heavy = emu.counters()["heavy_clusters"]
    assert (heavy > 20) if sums_cap < 100 else (heavy < 20), heavy


@pytest.mark.parametrize("overrides,read_lengths", [
    (dict(scatter_repeats=1), (150, 150)),
    (dict(dodgy_alignment_score=-1, mapq_threshold=20, keep_unaligned=0), (150, 150)),
    (dict(clip_semialigned=0, clip_overlapping=0, dodgy_alignment_score=255, mapq_threshold=3), (100, 100)),
    (dict(scatter_repeats=1, keep_unaligned=0), (150, 0)),
])
def test_lean_template_stage_with_other_options(oracle, emulib, overrides, read_lengths):
    """template_lean.h (what k_plan_rescue and k_select run) away from the defaults: --scatter-repeats (the tie picked by cluster id),
    --dodgy-alignment-score Unaligned / a forced value, --mapq-threshold, --keep-unaligned discard, clippers off, single-ended data; on a
    repeat-rich reference so that ties, rescued ties and low MAPQs occur; every record against the oracle, and the general form of
    template.h gives the same records"""
    from isaac_aligner_amd import synth
    g = synth.make_human_like_genome(1_200_000, seed=31)
    contigs = [bytes(c.numpy()) for c in g.contigs]
    bcl = synth.make_read_pairs(g, 1200, read_lengths[0], seed=32, avoid_gaps=True, subst_rate=0.01)[0].numpy()
    if not read_lengths[1]:
        bcl = np.ascontiguousarray(bcl[:, :read_lengths[0]])
    n = len(bcl)
    p = options.default_params(*read_lengths, **overrides)
    ref = oracle.reference(contigs)
    ref.build_index()
    matches, hits = ref.find_matches(p, bcl, n)
    otls = ref.determine_tls(p, bcl, matches, hits)
    orec, ocig, _ = ref.select(p, bcl, matches, otls, hits, n_clusters_hint=n)
    for lean in (1, 0):
        emu = hostemu_lib.Emu(emulib, p, contigs, hits)
        emu.set_matches(matches, n)
        emulib.emu_set_lean(emu.h, C.c_int(lean))
        etls = emu.determine_tls(bcl, n)
        assert otls.astuple() == etls.astuple()
        erec, ecig = emu.select(bcl, n, etls, n_reads=p.n_reads)
        assert not (erec["reserved"] & 5).any()
        erec = erec[(erec["reserved"] & 2) == 0]          # templates the reference does not store (--keep-unaligned discard): the oracle leaves them out
        assert not compare_records(orec, ocig, erec, ecig), lean


def test_gap_realigner_device_code_against_the_oracle():
    """realign.h (what the BAM stage's realign kernel runs per fragment) compiled for the CPU.  First in the reference test's own configuration
    (--realign-vigorously 1, round 6): the 62 cases of lib/build/cppunit/testGapRealigner.cpp give the positions, CIGARs and edit distances the
    reference asserts.  Then against oracle/realign.cpp, which those cases pin: the cases' inputs with and without --realign-vigorously, with the
    semialigned clipper off and on, with the test's costs and the BinSorter's (3, 4, 0); and random reads with planted indels against gap sets
    that hold the true gaps and decoys, both modes"""
    import ctypes as C
    import json
    lib = hostemu_lib.load()
    o = oracle_lib.load()
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gap_realigner.json")))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}

    def emu(case, realigner):
        bcl = np.array([(code[b] | 0x20) if b != "N" else 0 for b in case["read_bases"]], np.uint8)
        contig = case["contig"].encode()
        cigar = np.array(case["cigar"], np.uint32)
        gp = np.array([x[0] for x in case["gaps"]], np.int64); gl = np.array([x[1] for x in case["gaps"]], np.int32)
        pos, ncig, ed, obs = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        out = np.zeros(4096, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib.emu_realign_case(contig, C.c_uint64(len(contig)), p(bcl), C.c_uint32(len(bcl)), C.c_uint64(case["f_strand_position"]), p(cigar), C.c_uint32(len(cigar)),
                                  C.c_uint32(case["observed_length"]), C.c_uint32(case["edit_distance"]), C.c_uint32(case["low_clipped"]), C.c_uint32(case["high_clipped"]),
                                  p(gp), p(gl), C.c_uint32(len(gp)), C.c_uint32(case["mismatch_cost"]), C.c_uint32(case["gap_open_cost"]), C.c_uint32(realigner["gap_extend_cost"]),
                                  C.c_int(int(realigner["dodgy"])), C.c_int(int(realigner["clip_semialigned"])), C.c_int(int(realigner["vigorous"])), C.c_uint64(case["bin_start"]),
                                  C.c_int64(-1 if case["bin_end"] is None else case["bin_end"]), C.byref(pos), p(out), C.byref(ncig), C.byref(ed), C.byref(obs))
        assert rc == 0
        return {"position": pos.value, "cigar": oracle_lib.cigar_string(out[:ncig.value]), "edit_distance": ed.value, "observed_length": obs.value}

    # the reference's asserted answers from the device code
    assert g["realigner"]["vigorous"] is True
    for k, case in enumerate(g["cases"]):
        e, got = case["expected"], emu(case, g["realigner"])
        assert got["position"] == e["realignedPos_"] and got["edit_distance"] == e["realignedEditDistance_"], (k, got, e)
        if "realignedCigar_" in e:
            assert got["cigar"] == e["realignedCigar_"], (k, got, e)
    changed = 0
    for clip, vigorous in ((False, False), (True, False), (False, True), (True, True)):
        for costs in (None, (3, 4)):
            realigner = dict(g["realigner"], vigorous=vigorous, clip_semialigned=clip)
            for k, case in enumerate(g["cases"]):
                c = dict(case)
                if costs:
                    c["mismatch_cost"], c["gap_open_cost"] = costs
                want = o.realign_case(c, realigner)
                want.pop("overlaps")
                got = emu(c, realigner)
                assert got == want, (k, clip, costs, got, want)
                changed += got["cigar"] != oracle_lib.cigar_string(case["cigar"])
    assert changed > 100
    # random cases: a read copied from the contig with substitutions and one or two indels, aligned ungapped where it was taken from
    rng = np.random.default_rng(8)
    n_changed = 0
    for trial in range(600):
        realigner = dict(g["realigner"], vigorous=trial >= 400, clip_semialigned=True)
        contig = "".join("ACGT"[x] for x in rng.integers(0, 4, 400))
        start, L = int(rng.integers(20, 150)), 100
        read = list(contig[start:start + L + 30])
        gaps, at = [], int(rng.integers(15, 40))
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(1, 6))
            if rng.random() < 0.5:                      # deletion from the read: the reference keeps n bases the read lacks
                gaps.append((start + at, n)); del read[at:at + n]
            else:                                       # insertion into the read
                gaps.append((start + at, -n)); read[at:at] = list("ACGT"[x] for x in rng.integers(0, 4, n))
            at += int(rng.integers(15, 35))
        read = read[:L]
        for i in rng.integers(0, L, int(rng.integers(0, 3))):
            read[i] = "ACGT"[(code[read[i]] + 1) % 4]
        # decoys and the true gaps shifted into place (gap positions are those of the ungapped read's frame only for the first gap; the rest is what a real run has too: gaps of other reads)
        for _ in range(int(rng.integers(0, 4)) if trial < 500 else int(rng.integers(8, 16))):          # (the last hundred: more than ten gaps in reach)
            gaps.append((int(rng.integers(start, start + L)), int(rng.choice([-3, -1, 1, 2, 4]))))
        ed = sum(1 for i in range(L) if read[i] != contig[start + i])
        case = {"read_bases": "".join(read), "contig": contig, "f_strand_position": start, "cigar": [(L << 4) | 0], "observed_length": L, "edit_distance": ed, "low_clipped": 0, "high_clipped": 0,
                "gaps": gaps, "mismatch_cost": 3, "gap_open_cost": 4, "bin_start": 0, "bin_end": None}
        want = o.realign_case(case, realigner)
        want.pop("overlaps")
        got = emu(case, realigner)
        assert got == want, (trial, case, got, want)
        n_changed += got["cigar"] != "100M"
    assert n_changed > 150
