"""BASELINE.json's full-size configurations under pytest (-m gpu): the GRCh38-sized reference (3.1 Gbp, 25 contigs, 2.9 G index
entries), 2x150 (configuration 2) and 2x250 with 5 % indel reads (configuration 4), every record and BAM byte against the oracle; and
the index itself checked against witnesses that do not use it: the oracle's own builder on a 100 Mbp human-like genome, and at
3.1 Gbp order / mask / count invariants plus a brute-force scan of the reference (tests/gpucheck) for sampled k-mers."""
import json
import os
import time

import numpy as np
import pytest

import gpucheck_lib
import oracle_lib
from isaac_aligner_amd import abi, options, synth
from parity_util import count_record_diffs, sort_matches

pytestmark = pytest.mark.gpu

GENOME_BASES = int(os.environ.get("ISAAC_SCALE_GENOME_BASES", 3_100_000_000))
PAIRS = int(os.environ.get("ISAAC_SCALE_PAIRS", 1_000_000))
SIGN = -(1 << 63)          # k-mers are compared as unsigned numbers: flipping the top bit makes int64 order the same order


def flipped(u):
    """the int64 whose order among int64s is the order of the unsigned 64-bit value u among unsigned values"""
    v = (u ^ (1 << 63)) & ((1 << 64) - 1)
    return v - (1 << 64) if v >= (1 << 63) else v


def rank_in_table(torch, kmers, n, flipped_queries, right=False, step=1 << 27):
    """number of table entries below (or not above) every query, chunk by chunk (one searchsorted over 2.9 G elements wants a 23 GB temporary)"""
    total = torch.zeros_like(flipped_queries)
    for a in range(0, n, step):
        kk = kmers[a:min(n, a + step)] ^ SIGN
        total += torch.searchsorted(kk, flipped_queries, right=right)
        del kk
    return total


@pytest.fixture(scope="module")
def human(torch):
    """the reference and its index, built once for the module (about 30 s)"""
    from isaac_aligner_amd import gpu
    dev = torch.device("cuda", 0)
    genome = synth.make_human_like_genome(GENOME_BASES, seed=3, device=dev)
    torch.cuda.empty_cache()
    al = gpu.Aligner(options.default_params(150, 150), 0, genome)
    n = al.build_index(repeat_threshold=1000, annotate_neighbors=True)
    state = {"genome": genome, "al": al, "n": n, "oracle_ref": None}
    yield state
    al.close()


def oracle_reference(human, oracle):
    """the oracle's view of the same reference; the table is handed over (its own correctness is the business of the index tests below)"""
    if human["oracle_ref"] is None:
        ref = oracle.reference([c.cpu().numpy().tobytes() for c in human["genome"].contigs])
        ref.set_index(human["al"].get_index())
        human["oracle_ref"] = ref
    return human["oracle_ref"]


def align_and_compare(torch, oracle, human, L, tile, **read_kw):
    """find -> template length statistics -> select -> packed CIGARs -> BAM records on the device, the same on the oracle, compared in full"""
    al, genome = human["al"], human["genome"]
    p = options.default_params(L, L)
    al.set_params(p)
    bcl = synth.make_read_pairs(genome, PAIRS, L, seed=4000 + L, device=al.device, avoid_gaps=True, **read_kw)[0]
    matches, offsets, hits = al.find_matches(bcl, tile=tile)
    al.set_loaded_contigs(hits)
    tls = al.determine_tls(bcl, matches, offsets, tile=tile)
    records, cigars = al.select(bcl, matches, offsets, tls, tile=tile)
    al.synchronize()
    packed, n_words = al.compact_cigars(records, cigars)
    counters = al.counters()
    ref = oracle_reference(human, oracle)
    host_bcl = bcl.cpu().numpy()
    cores = os.cpu_count() or 1
    om, ohits = ref.find_matches(p, host_bcl, PAIRS, tile=tile, n_threads=min(cores, 64))
    gm = matches.cpu().numpy().view(np.uint64).reshape(-1, 2)
    gm = np.rec.fromarrays([gm[:, 0], gm[:, 1]], dtype=oracle_lib.MATCH_DTYPE)
    a, b = sort_matches(om), sort_matches(gm)                                # the oracle also lists NoMatch records; they are dropped here
    assert len(a) == len(b) and (a["seed_id"] == b["seed_id"]).all() and (a["location"] == b["location"]).all() and (ohits == hits).all()
    otls = ref.determine_tls(p, host_bcl, om, ohits, tile=tile)
    assert otls.astuple() == tls.astuple()
    orec, ocig, _ = ref.select(p, host_bcl, om, otls, ohits, tile=tile, n_threads=cores, n_clusters_hint=PAIRS)
    grec = records.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1)
    gcig = packed.cpu().numpy().view(np.uint32)
    n_diff, text = count_record_diffs(orec, ocig, grec, gcig)
    assert n_diff == 0, "\n".join(text)
    assert not (grec["reserved"] & 5).any()                                   # no capacity was exceeded
    prefix = "SCALE:1:%d:" % tile
    gbam = al.bam_records([(bcl, records, packed, prefix)])[0].cpu().numpy().tobytes()
    obam = oracle.bam_records([(host_bcl, orec, ocig, prefix)], [L, L], forced_dodgy_alignment_score=p.dodgy_alignment_score & 0xff)[0]
    assert gbam == obam
    # the same with the reference's defaults for the BAM stage: --mark-duplicates 1 --keep-duplicates 1 --realign-gaps sample
    gbam, n, un = al.bam_records([(bcl, records, packed, prefix)], mark_duplicates=True, keep_duplicates=True, realign_gaps=True, tls=tls)
    obam, on, oun = oracle.bam_records([(host_bcl, orec, ocig, prefix)], [L, L], forced_dodgy_alignment_score=p.dodgy_alignment_score & 0xff, mark_duplicates=True, keep_duplicates=True,
                                       realign_gaps=True, clip_semialigned=True, reference=ref, tls=otls)
    assert (n, un) == (on, oun) and gbam.cpu().numpy().tobytes() == obam
    return counters, grec


def test_configuration_2_grch38_2x150(torch, oracle, human):
    counters, rec = align_and_compare(torch, oracle, human, 150, tile=7)
    aligned = ((rec["flags"] & 2) == 0).mean()
    assert aligned > 0.97 and counters["rescue_calls"] > PAIRS // 2 and counters["bsw_jobs"] > PAIRS // 10


def test_configuration_4_grch38_2x250_indels(torch, oracle, human):
    counters, rec = align_and_compare(torch, oracle, human, 250, tile=9, indel_read_fraction=0.05, indel_max=10)
    gapped = (rec["gap_count"] > 0).mean()
    assert gapped > 0.03 and counters["bsw_jobs"] > PAIRS // 10


def test_compact_cigars_retry_with_a_pool_that_is_too_small(torch, human):
    """ISAAC_GPU_ECAPACITY must leave the records as they were, so that the documented retry packs the same CIGARs"""
    al, genome = human["al"], human["genome"]
    p = options.default_params(150, 150)
    al.set_params(p)
    bcl = synth.make_read_pairs(genome, 20000, 150, seed=77, device=al.device, avoid_gaps=True)[0]
    records, cigars = al.align_tile(bcl, tile=3)
    al.synchronize()
    before = records.clone()
    want, n_want = al.compact_cigars(records.clone(), cigars)
    small = torch.empty(100, dtype=torch.int32, device=al.device)
    from isaac_aligner_amd import gpu
    import ctypes as C
    n = C.c_uint64()
    rc = al.lib.isaac_gpu_compact_cigars(al.h, C.c_void_p(records.data_ptr()), C.c_uint64(records.shape[0]), C.c_void_p(cigars.data_ptr()), C.c_void_p(small.data_ptr()),
                                         C.c_uint64(small.numel()), C.byref(n))
    assert rc == 4 and n.value == n_want and (records == before).all()
    got, n_got = al.compact_cigars(records, cigars, out=small)              # the wrapper's retry path: allocates what the first call reported
    assert n_got == n_want and (got == want).all()
    assert isinstance(gpu.IsaacGpuError("x"), RuntimeError)


def test_bam_stage_defaults_on_deep_coverage(torch, oracle):
    """Duplicate marking and gap realignment where they have work to do, at the size of a launch: half a million pairs of a sample that
    carries an indel every ~400 bases over a 2 Mbp reference (50-fold coverage, so every indel is shown by tens of reads and crossed near
    a read end by as many; a tenth of the fragments sequenced twice), isaac_gpu_bam_records with the reference's defaults against the oracle,
    byte for byte."""
    from isaac_aligner_amd import gpu
    L, n_pairs = 100, int(os.environ.get("ISAAC_SCALE_DEEP_PAIRS", 500_000))
    rng = np.random.default_rng(23)
    genome = synth.make_genome(2_000_000, seed=67, n_contigs=3, repeat_families=True)
    contigs = [bytes(c.numpy()) for c in genome]
    sample = synth.make_sample_with_indels(genome, rng)
    p = options.default_params(L, L)
    al = gpu.Aligner(p, 0, contigs)
    al.build_index()
    ref = oracle.reference(contigs)
    ref.set_index(al.get_index())
    host_bcl = synth.make_read_pairs(sample, n_pairs, L, seed=91, indel_read_fraction=0.0, subst_rate=0.004)[0].numpy()
    twice = n_pairs // 10
    host_bcl[n_pairs - twice:] = host_bcl[rng.integers(0, n_pairs - twice, twice)]
    bcl = torch.from_numpy(host_bcl).cuda()
    matches, offsets, hits = al.find_matches(bcl, tile=2)
    al.set_loaded_contigs(hits)
    tls = al.determine_tls(bcl, matches, offsets, tile=2)
    records, cigars = al.select(bcl, matches, offsets, tls, tile=2)
    al.synchronize()
    packed, _ = al.compact_cigars(records, cigars)
    cores = os.cpu_count() or 1
    om, ohits = ref.find_matches(p, host_bcl, n_pairs, tile=2, n_threads=min(cores, 64))
    otls = ref.determine_tls(p, host_bcl, om, ohits, tile=2)
    assert otls.astuple() == tls.astuple()
    orec, ocig, _ = ref.select(p, host_bcl, om, otls, ohits, tile=2, n_threads=cores, n_clusters_hint=n_pairs)
    grec = records.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1)
    n_diff, text = count_record_diffs(orec, ocig, grec, packed.cpu().numpy().view(np.uint32))
    assert n_diff == 0, "\n".join(text)
    prefix = "DEEP:1:2:"
    dodgy = p.dodgy_alignment_score & 0xff
    plain = oracle.bam_records([(host_bcl, orec, ocig, prefix)], [L, L], forced_dodgy_alignment_score=dodgy)[0]
    for mark, keep in ((True, True), (True, False)):
        got, n, un = al.bam_records([(bcl, records, packed, prefix)], mark_duplicates=mark, keep_duplicates=keep, realign_gaps=True, tls=tls)
        want, wn, wun = oracle.bam_records([(host_bcl, orec, ocig, prefix)], [L, L], forced_dodgy_alignment_score=dodgy, mark_duplicates=mark, keep_duplicates=keep,
                                           realign_gaps=True, clip_semialigned=True, reference=ref, tls=otls)
        assert (n, un) == (wn, wun)
        assert got.cpu().numpy().tobytes() == want
        if keep:
            from isaac_aligner_amd import bam
            a, b = bam.parse_records(plain), bam.parse_records(want)
            by_name = {(r["name"], r["flag"] & 0xc0): r for r in a}
            n_dup = sum(1 for r in b if r["flag"] & 0x400)
            n_moved = sum(1 for r in b if list(by_name[(r["name"], r["flag"] & 0xc0)]["cigar"]) != list(r["cigar"]))
            assert n_dup > twice and n_moved > n_pairs // 500, (n_dup, n_moved)
    al.close()


# ---- the index ------------------------------------------------------------------------------------------------------------------------

def test_index_against_the_oracle_builder_on_100_mbp(torch, oracle):
    """isaac_gpu_build_index against ReferenceSorter + NeighborsFinder as the oracle restates them (70 permutations, sort, compare inside
    equal-prefix blocks), entry for entry including every neighbour bit, on a human-like genome 250 times the size of the small tests'"""
    from isaac_aligner_amd import gpu
    n_bases = int(os.environ.get("ISAAC_SCALE_ORACLE_INDEX_BASES", 100_000_000))
    dev = torch.device("cuda", 0)
    genome = synth.make_human_like_genome(n_bases, seed=11, device=dev)
    al = gpu.Aligner(options.default_params(150, 150), 0, genome)
    n = al.build_index()
    got = al.get_index()
    cuts = al.mask_offsets()
    al.close()
    ref = oracle.reference([c.cpu().numpy().tobytes() for c in genome.contigs])
    t0 = time.time()
    want = ref.build_index(repeat_threshold=1000, annotate_neighbors=True, n_threads=os.cpu_count() or 1)
    print("oracle builder: %d entries in %.1f s" % (len(want), time.time() - t0))
    assert n == len(want) == len(got)
    assert (got["kmer"] == want["kmer"]).all()
    assert (got["position"] >> np.uint64(1) == want["position"] >> np.uint64(1)).all()
    assert ((got["position"] & np.uint64(1)) == (want["position"] & np.uint64(1))).all()
    flagged = int((want["position"] & np.uint64(1)).sum())
    too_many = int(((want["position"] >> np.uint64(1)) == 0).sum())
    assert flagged > n // 100 and too_many > 0                                 # the genome does exercise both
    assert [int(c) for c in cuts] == [int(np.searchsorted(want["kmer"] >> np.uint64(58), m)) for m in range(64)] + [n]


def valid_window_count(torch, genome):
    """forward 32-mers of the reference: windows of 32 ACGT bases that stay inside their contig (ReferenceSorter.cpp:105-177)"""
    total = 0
    chunk = 1 << 28
    for c in genome.contigs:
        n = c.numel()
        run = 0                                                                # ACGT bases immediately before the chunk
        for a in range(0, n, chunk):
            part = c[a:a + chunk]
            ok = (part == 65) | (part == 67) | (part == 71) | (part == 84)
            # length of the run of valid bases ending at every position: position - last invalid position
            idx = torch.arange(a, a + part.numel(), device=part.device)
            last_bad = torch.where(~ok, idx, torch.full_like(idx, -1))
            last_bad = torch.cummax(last_bad, 0)[0]
            last_bad = torch.where(last_bad < 0, torch.full_like(last_bad, a - run - 1), last_bad)
            length = idx - last_bad
            total += int((length >= 32).sum())
            tail_bad = int(last_bad[-1])
            run = a + part.numel() - 1 - tail_bad
            del idx, last_bad, length, ok
    return total


def test_index_invariants_at_full_size(torch, human):
    """what must hold for the 2.9 G-entry table whatever built it: global (k-mer, position) order, the mask cuts, entry count against the
    reference's own count of forward 32-mers, TooManyMatch entries single"""
    al, genome, n = human["al"], human["genome"], human["n"]
    entries = al.index_tensors()
    kmers, positions = entries[:, 0], entries[:, 1]                       # (strided views of the interleaved table)
    assert kmers.numel() == n == positions.numel()
    cuts = al.mask_offsets()
    assert cuts[0] == 0 and cuts[-1] == n
    step = 1 << 27
    n_too_many = 0
    for a in range(0, n, step):
        b = min(n, a + step + 1)
        k = kmers[a:b] ^ SIGN
        pz = positions[a:b]
        assert bool((k[1:] >= k[:-1]).all()), "k-mers out of order in [%d, %d)" % (a, b)
        same = k[1:] == k[:-1]
        # equal k-mers: forward occurrences in reference order (ReferencePosition values ascending, neighbour bit aside), never a TooManyMatch entry
        assert bool(((pz[1:] >> 1) > (pz[:-1] >> 1))[same].all())
        tm = (pz >> 1) == 0
        assert not bool((tm[1:] & same).any()) and not bool((tm[:-1] & same).any())
        n_too_many += int(tm[:min(n, a + step) - a].sum())
        del k, pz, same, tm
    bounds = torch.tensor([flipped(m << 58) for m in range(64)], dtype=torch.int64, device=kmers.device)
    got_cuts = rank_in_table(torch, kmers, n, bounds).cpu().numpy()           # entries before mask m = entries whose k-mer is below m << 58
    assert [int(c) for c in cuts[:64]] == [int(c) for c in got_cuts]
    n_windows = valid_window_count(torch, genome)
    assert n_too_many > 0 and n <= n_windows
    # every forward 32-mer is either an entry of its own or one of > 1000 occurrences folded into a TooManyMatch entry
    human["n_windows"], human["n_too_many"] = n_windows, n_too_many
    print("entries %d, forward 32-mers %d, TooManyMatch entries %d" % (n, n_windows, n_too_many))


def pack_kmers(torch, genome, starts):
    """packed 32-mers (int64, bit pattern of the uint64) at global positions `starts` + validity (all ACGT, inside one contig)"""
    dev = genome.bases.device
    idx = starts.unsqueeze(1) + torch.arange(32, device=dev).unsqueeze(0)
    b = genome.padded[idx.clamp(max=genome.padded.numel() - 1)]
    code = torch.full_like(b, 4)
    code[b == 65], code[b == 67], code[b == 71], code[b == 84] = 0, 1, 2, 3
    offsets = torch.tensor(genome.offsets, dtype=torch.long, device=dev)
    contig = torch.searchsorted(offsets, starts, right=True) - 1
    valid = (code < 4).all(1) & (starts + 32 <= offsets[contig + 1])
    k = torch.zeros(len(starts), dtype=torch.int64, device=dev)
    for i in range(32):
        k = (k << 2) | (code[:, i].long() & 3)
    return k, valid, contig


def test_index_entries_against_a_brute_force_scan(torch, human):
    """Sampled 32-mers -- drawn from reference positions, from table entries, and from the TooManyMatch entries -- are counted and searched
    for 1..4-mismatch neighbours by scanning all of the reference on both strands (tests/gpucheck: no table, no sorting, no shared code),
    and the table must say exactly that: a single TooManyMatch entry above 1000 occurrences, else one entry per forward occurrence at the
    right places, all carrying the neighbour bit the scan found."""
    al, genome, n = human["al"], human["genome"], human["n"]
    dev = genome.bases.device
    entries = al.index_tensors()
    kmers, positions = entries[:, 0], entries[:, 1]
    g = torch.Generator(device=dev).manual_seed(99)
    n_positions = int(os.environ.get("ISAAC_SCALE_SCAN_SAMPLES", 12288))
    total = genome.offsets[-1]
    starts = (torch.rand(2 * n_positions, generator=g, device=dev, dtype=torch.float64) * (total - 32)).long()
    k, valid, _ = pack_kmers(torch, genome, starts)
    from_positions = k[valid][:n_positions]
    pick = (torch.rand(n_positions // 6, generator=g, device=dev, dtype=torch.float64) * n).long()
    from_table = kmers[pick]
    tm_at = torch.nonzero((positions >> 1) == 0).flatten() if n < (1 << 31) else torch.cat(
        [torch.nonzero((positions[a:a + (1 << 30)] >> 1) == 0).flatten() + a for a in range(0, n, 1 << 30)])
    assert tm_at.numel() > 0
    tm_pick = tm_at[(torch.rand(min(n_positions // 6, tm_at.numel()), generator=g, device=dev, dtype=torch.float64) * tm_at.numel()).long()]
    from_too_many = kmers[tm_pick]
    q = torch.unique(torch.cat([from_positions, from_table, from_too_many]))
    qh = q.cpu().numpy().view(np.uint64)
    rc = gpucheck_lib.reverse_complement(qh)
    t0 = time.time()
    count, possum, near = gpucheck_lib.kmer_scan(genome.padded, genome.offsets, np.concatenate([qh, rc]))
    print("brute-force scan of %d queries: %.1f s" % (2 * len(qh), time.time() - t0))
    m = len(qh)
    fwd, rev = count[:m].astype(np.int64), count[m:].astype(np.int64)
    occurrences = fwd + rev                                                   # both strands, as ReferenceSorter counts (a palindrome counts twice)
    has_neighbor = near[:m] | near[m:]
    # what the table holds for the same k-mers
    flipped_q = q ^ SIGN
    lo_all = rank_in_table(torch, kmers, n, flipped_q)
    hi_all = rank_in_table(torch, kmers, n, flipped_q, right=True)
    run = (hi_all - lo_all).cpu().numpy()
    lo_h = lo_all.cpu().numpy()
    repeat = occurrences > 1000
    # a k-mer drawn from a reference position or a table entry occurs on the forward strand; one drawn from a TooManyMatch entry more than 1000 times
    assert (fwd >= 1).all()
    expected_run = np.where(repeat, 1, fwd)
    bad = np.nonzero(run != expected_run)[0]
    assert not len(bad), "entries per k-mer differ from the scan for %d of %d k-mers, e.g. k-mer %016x: table %d, scan fwd %d rev %d" % (
        len(bad), m, int(qh[bad[0]]), int(run[bad[0]]), int(fwd[bad[0]]), int(rev[bad[0]]))
    # the entries themselves
    seg = torch.repeat_interleave(torch.arange(m, device=dev), torch.from_numpy(run).to(dev))
    first = torch.from_numpy(lo_h).to(dev)
    csum = torch.cumsum(torch.from_numpy(run).to(dev), 0) - torch.from_numpy(run).to(dev)
    at = first[seg] + (torch.arange(seg.numel(), device=dev) - csum[seg])
    entry_pos = positions[at]
    assert bool((kmers[at] == q[seg]).all())
    is_tm = ((entry_pos >> 1) == 0).cpu().numpy()
    seg_h = seg.cpu().numpy()
    assert (is_tm == repeat[seg_h]).all()                                    # TooManyMatch exactly where the scan counted more than 1000
    sums = np.zeros(m, np.uint64)
    np.add.at(sums, seg_h[~is_tm], (entry_pos.cpu().numpy().view(np.uint64)[~is_tm] >> np.uint64(1)) << np.uint64(1))
    assert (sums[~repeat] == possum[:m][~repeat]).all()                      # at the places where the scan saw the k-mer
    bits = (entry_pos & 1).cpu().numpy().astype(bool)
    first_entry = np.concatenate([[True], seg_h[1:] != seg_h[:-1]])
    kmer_bits = np.zeros(m, bool); kmer_bits[seg_h[first_entry & ~is_tm]] = bits[first_entry & ~is_tm]
    print("k-mers: table 1 / scan 1: %d, table 0 / scan 1: %d, table 1 / scan 0: %d, neither: %d" % (
        int((kmer_bits & has_neighbor & ~repeat).sum()), int((~kmer_bits & has_neighbor & ~repeat).sum()), int((kmer_bits & ~has_neighbor & ~repeat).sum()),
        int((~kmer_bits & ~has_neighbor & ~repeat).sum())))
    wrong = np.nonzero((bits != has_neighbor[seg_h]) & ~is_tm)[0]
    assert not len(wrong), "%d of %d entries carry a neighbour bit the scan contradicts, e.g. k-mer %016x: table %d, scan %d" % (
        len(wrong), len(bits), int(qh[seg_h[wrong[0]]]), int(bits[wrong[0]]), int(has_neighbor[seg_h[wrong[0]]]))
    assert has_neighbor.any() and not has_neighbor.all() and repeat.any()
    print("k-mers checked %d (entries %d): %d with neighbours, %d repeats" % (m, len(bits), int(has_neighbor.sum()), int(repeat.sum())))


def test_isaac_align_on_the_full_size_reference(torch, oracle, human, tmp_path):
    """bin/isaac-align at real size: the 3.1 Gbp reference as isaac-sort-reference leaves it (FASTA + 64 mask files, 47 GB, read back by
    isaac_gpu_load_sorted_reference), three lanes of 5.1 M pairs of 2x150 as FASTQ files -- without --clusters-at-a-time a lane is one load cut into
    two tiles (5 M + 0.1 M clusters: 40 M / 8 seeds per tile) -- aligned with the reference's defaults (duplicates marked, gaps realigned,
    --bam-gzip-level 1 on the device), binned by contig into host memory and built bin by bin.  sorted.bam must inflate to the header and record
    stream of the oracle chain on the same reads (lookup, statistics per lane, selection, duplicates, realignment, order, records), and the .bai must
    be what the oracle's BamIndexer makes of those records and the file's own BGZF blocks."""
    import shutil
    import subprocess
    import tempfile
    import zlib
    from isaac_aligner_amd import bam, build, sorted_reference as sr
    al, genome = human["al"], human["genome"]
    L = 150
    pairs_per_lane = int(os.environ.get("ISAAC_SCALE_CLI_PAIRS_PER_LANE", 5_100_000))
    lanes = (1, 2, 5)
    p = options.default_params(L, L)
    al.set_params(p)
    work = tempfile.mkdtemp(prefix="isaac_scale_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path))
    try:
        ref_dir, calls = os.path.join(work, "ref"), os.path.join(work, "calls")
        os.makedirs(ref_dir); os.makedirs(calls)
        fasta = os.path.join(ref_dir, "genome.fa")
        contigs, position = [], 0
        for i, (offset, size, bases, acgt) in enumerate(synth.write_fasta(fasta, genome.contigs)):
            m = sr.Contig()
            m.genomic_position, m.index, m.karyotype_index, m.name, m.file = position, i, i, b"chr%d" % (i + 1), fasta.encode()
            m.offset, m.size, m.total_bases, m.acgt_bases = offset, size, bases, acgt
            position += bases
            contigs.append(m)
        al.save_sorted_reference(ref_dir, "genome.fa", contigs)
        # the reads: per lane two FASTQ files; the BCL bytes the oracle gets are the ones the text was written from (the text -> BCL conversion has
        # tests of its own; a prefix of it is checked here)
        lane_bcl = {}
        for lane in lanes:
            parts = [synth.make_read_pairs(genome, min(1_000_000, pairs_per_lane - first), L, seed=9000 + 100 * lane + first // 1_000_000, device=al.device, avoid_gaps=True)[0].cpu()
                     for first in range(0, pairs_per_lane, 1_000_000)]
            bcl = torch.cat(parts).numpy()
            lane_bcl[lane] = bcl
            files = [open(os.path.join(calls, "lane%d_read%d.fastq" % (lane, r + 1)), "wb") for r in range(2)]
            for first in range(0, len(bcl), 1_000_000):
                synth.write_fastq(files, bcl[first:first + 1_000_000], L, name_prefix=b"M1:7:FCSCALE:%d:1101:" % lane, first_index=first)
            for f in files:
                f.close()
        with open(os.path.join(calls, "lane1_read1.fastq"), "rb") as f:
            text = f.read(2000 * (2 * L + 40))
        text = text[:text.rfind(b"\n@") + 1]
        rc, back, n_back, _, _ = oracle.fastq_to_bcl(text, L, max_clusters=4000)
        assert rc == 0 and n_back > 1000 and (back[:n_back, :L] == lane_bcl[1][:n_back, :L]).all()
        out = os.path.join(work, "Aligned")
        args = [build.build_host(), "-r", os.path.join(ref_dir, "sorted-reference.xml"), "-b", calls, "--base-calls-format", "fastq", "-o", out, "--use-bases-mask", "y*,y*", "-j", "64"]
        t0 = time.time()
        r = subprocess.run(args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        print("isaac-align: %.1f s\n%s" % (time.time() - t0, "\n".join(l for l in r.stderr.splitlines() if "done in" in l or "timing" in l or "tile(s)" in l)))
        assert "%d clusters in 6 tile(s)" % (3 * pairs_per_lane) in r.stderr
        timing = json.loads([l for l in r.stderr.splitlines() if "timing {" in l][-1].split("timing ", 1)[1])
        assert timing["overflow_clusters"] == 0 and timing["mapq_changed_by_host"] == 0
        # the bins the program made (about 4 M records each: the long contigs are cut), for the oracle's BAM stage and for the index's parts
        bin_ranges = [tuple(r_) for r_ in timing["bin_ranges"]]
        bin_cuts = [first for first, _ in bin_ranges if (first >> 1) & ((1 << 40) - 1)]
        assert timing["bin_cuts"] == len(bin_cuts)           # (none at this depth: 30.6 M records in bins of 4 M are 400 Mbp a bin, more than any contig; test_cli cuts)
        # ---- the oracle chain
        ref = oracle_reference(human, oracle)
        cores = os.cpu_count() or 1
        tile_max = 40_000_000 // p.n_seeds
        found, all_hits = [], np.zeros(len(genome.contigs), np.uint8)
        for lane_index, lane in enumerate(lanes):
            bcl, number = lane_bcl[lane], 1
            for first in range(0, len(bcl), tile_max):
                tile_bcl = np.ascontiguousarray(bcl[first:first + tile_max])
                index = (lane_index << 16) | (number - 1)              # the program's tile index: lane ordinal, tile ordinal in the lane
                om, hits = ref.find_matches(p, tile_bcl, len(tile_bcl), tile=index & 0xfff, n_threads=min(cores, 64))
                all_hits |= hits
                found.append((lane_index, lane, number, index, tile_bcl, om))
                number += 1
        host_tiles, tls_of_lane = [], {}
        for lane_index, lane, number, index, tile_bcl, om in found:
            tls = tls_of_lane.get(lane_index)
            if tls is None or not tls.stable:
                tls = tls_of_lane[lane_index] = ref.determine_tls(p, tile_bcl, om, all_hits, tile=index)
            orec, ocig, _ = ref.select(p, tile_bcl, om, tls, all_hits, tile=index, n_threads=cores, n_clusters_hint=len(tile_bcl))
            host_tiles.append((tile_bcl, orec, ocig, "FCSCALE:%d:%d:" % (lane, number), str(lane_index), tls))
        del found
        want, want_n, want_unaligned = oracle.bam_records(host_tiles, [L, L], forced_dodgy_alignment_score=p.dodgy_alignment_score & 0xff, mark_duplicates=True, keep_duplicates=True,
                                                          realign_gaps=True, reference=ref, bin_cuts=bin_cuts)
        assert want_n == 2 * 3 * pairs_per_lane
        sq = [("chr%d" % (i + 1), int(c.total_bases), "", fasta, "") for i, c in enumerate(contigs)]
        header = oracle.bam_header(" ".join(args), "isaac_aligner_amd-0.3", sq, header_lines=["@RG\tID:%d\tPL:ILLUMINA\tSM:default\tPU:FCSCALE:%d:none" % (k, lane) for k, lane in enumerate(lanes)])
        # ---- the file: inflated block by block against header + records; its blocks' places for the index
        path = os.path.join(out, "Projects", "default", "default", "sorted.bam")
        data = np.fromfile(path, np.uint8)
        expected_total = len(header) + len(want)
        blocks, at, raw_at = [], 0, 0
        view = memoryview(data)
        hv, wv = memoryview(header), memoryview(want)
        while at < len(data):
            assert bytes(view[at:at + 4]) == b"\x1f\x8b\x08\x04" and bytes(view[at + 12:at + 14]) == b"BC"
            size = int.from_bytes(bytes(view[at + 16:at + 18]), "little") + 1
            raw = zlib.decompress(view[at + 18:at + size - 8], -15)
            assert int.from_bytes(bytes(view[at + size - 4:at + size]), "little") == len(raw)
            for piece_at, piece in ((raw_at, raw),):
                end = piece_at + len(piece)
                if end <= len(header):
                    assert piece == hv[piece_at:end]
                elif piece_at >= len(header):
                    assert piece == wv[piece_at - len(header):end - len(header)], "records differ in the block at %d" % at
                else:
                    assert piece == bytes(hv[piece_at:]) + bytes(wv[:end - len(header)])
            blocks.append((at, size, raw_at))
            raw_at += len(raw); at += size
        assert raw_at == expected_total and blocks[-1][1] == 28
        start_of = {b[2]: b[0] for b in blocks[:-1]}
        start_of[raw_at] = blocks[-1][0]
        from test_cli import split_by_bins
        cuts = split_by_bins(want, want_unaligned, bin_ranges)          # a BGZF run per bin and contig
        parts, file_at = [], len(header)
        for off, size in cuts:
            parts.append((off, size, bytes(view[start_of[file_at]:start_of[file_at + size]])))       # every bin starts a block of its own
            file_at += size
        header_bgzf = start_of[len(header)]
        bai = open(path + ".bai", "rb").read()
        assert bai == oracle.bam_index(want, parts, len(contigs), header_bgzf)
    finally:
        shutil.rmtree(work, ignore_errors=True)
