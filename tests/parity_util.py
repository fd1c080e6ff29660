"""helpers shared by the parity tests: synthetic inputs and record comparison"""
import numpy as np

from isaac_aligner_amd import abi, synth


def make_inputs(genome_bases=300000, n_pairs=2000, read_length=150, read_length2=None, seed=1, n_contigs=2, **kw):
    contigs = synth.make_genome(genome_bases, seed=seed, n_contigs=n_contigs)
    bcl, truth = synth.make_read_pairs(contigs, n_pairs, read_length, seed=seed + 1, read_length2=read_length2, **kw)
    return [bytes(c.numpy()) for c in contigs], bcl.numpy(), truth


def add_adapters(bcl, read_length, adapter="AGATCGGAAGAGC", fraction=0.3, seed=5, insert_range=(60, 145), adapter2=None):
    """Short-insert pairs: for `fraction` of the pairs (2 x read_length BCL bytes each) the fragment is cut to an insert shorter than the reads, so that both reads
    run into the sequencing adapter -- read 1 keeps its first `insert` bases, read 2 becomes the reverse complement of those, and either is followed by the adapter
    (as the sequencer reads it: the same text at the 3' end of both reads) and by random bases; qualities stay.  Returns (bcl copy, inserts: 0 = untouched)."""
    rng = np.random.default_rng(seed)
    out = bcl.copy()
    n, L = len(out), read_length
    inserts = np.zeros(n, np.int64)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    tails = [np.array([code[c] for c in a.encode()], np.uint8) for a in (adapter, adapter2 or adapter)]
    for i in np.nonzero(rng.random(n) < fraction)[0]:
        insert = int(rng.integers(insert_range[0], insert_range[1] + 1))
        r1 = out[i, :L].copy()
        if (r1[:insert] == 0).any():
            continue                                             # (an N inside the fragment: left alone)
        inserts[i] = insert
        fragment = r1[:insert] & 3
        for read, bases in ((0, fragment), (1, (3 - fragment)[::-1])):
            tail = np.concatenate([tails[read], rng.integers(0, 4, L, dtype=np.uint8)])
            seq = np.concatenate([bases, tail])[:L]
            q = out[i, read * L:(read + 1) * L] >> 2
            q = np.maximum(q, 1)                                 # (quality 0 would make the base an N)
            out[i, read * L:(read + 1) * L] = seq | (q << 2)
    return out, inserts


def compare_candidates(a, acig, b, bcig, limit=5):
    """a/b: CANDIDATE_DTYPE arrays in (cluster, read, list) order; returns list of textual differences"""
    diffs = []
    if len(a) != len(b):
        diffs.append("candidate count %d != %d" % (len(a), len(b)))
    for i in range(min(len(a), len(b))):
        x, y = a[i], b[i]
        same = all(x[f] == y[f] for f in x.dtype.names if f not in ("cigar_offset", "reserved", "first_seed_index"))
        cx = acig[x["cigar_offset"]:x["cigar_offset"] + x["cigar_length"]]
        cy = bcig[y["cigar_offset"]:y["cigar_offset"] + y["cigar_length"]]
        if not same or list(cx) != list(cy):
            diffs.append("candidate %d:\n  %s %s\n  %s %s" % (i, x, abi.cigar_string(cx), y, abi.cigar_string(cy)))
            if len(diffs) >= limit:
                break
    return diffs


def compare_records(a, acig, b, bcig, limit=5):
    """a/b: FRAGMENT_DTYPE arrays, one per read in cluster order; compares every io::FragmentHeader field, MAPQ and the CIGAR"""
    diffs = []
    if len(a) != len(b):
        diffs.append("record count %d != %d" % (len(a), len(b)))
    names = [f for f in a.dtype.names if f not in ("cigar_offset", "reserved")]
    n = min(len(a), len(b))
    eq = np.ones(n, bool)
    for f in names:
        eq &= a[f][:n] == b[f][:n]
    eq &= (a["reserved"][:n] >> 16) == (b["reserved"][:n] >> 16)          # the template's alignment score (bits 16-31; the low bits are diagnostics)
    for i in range(n):
        x, y = a[i], b[i]
        cx = acig[x["cigar_offset"]:x["cigar_offset"] + x["cigar_length"]]
        cy = bcig[y["cigar_offset"]:y["cigar_offset"] + y["cigar_length"]]
        if not eq[i] or list(cx) != list(cy):
            diffs.append("record %d:\n  %s %s\n  %s %s" % (i, x, abi.cigar_string(cx), y, abi.cigar_string(cy)))
            if len(diffs) >= limit:
                break
    return diffs


def sort_matches(m):
    """canonical order for comparing match sets: (cluster, location, seed, reverse); NoMatch records dropped"""
    m = m[m["location"] != abi.REFPOS_NOMATCH]
    cluster = (m["seed_id"] >> np.uint64(9)) & np.uint64(0x7fffffff)
    order = np.lexsort((m["seed_id"] & np.uint64(0x1ff), m["location"], cluster))
    return m[order]


# ---- lib/alignment/cppunit/testFragmentBuilder.cpp vectors (tests/golden/fragment_builder.json)
CANDIDATE_FIELD = {"contigId": "contig_id", "uniqueSeedCount": "unique_seed_count", "position": "position", "observedLength": "observed_length", "readIndex": "read_index",
                   "reverse": "reverse", "cigarOffset": "cigar_offset", "cigarLength": "cigar_length", "mismatchCount": "mismatch_count"}


def fragment_builder_params(g, repeat_threshold, device_limits=False):
    """the FragmentBuilder the test constructs (:92-94): 100 + 100 cycles, three 32-mer seeds per read at 0 / 32 / 64.
    device_limits: the test's repeat thresholds (123, 456) are above what the device library accepts (1..16, its fixed candidate
    capacity); no case has more than two candidates per read, so the largest accepted value gives the same lists."""
    if device_limits:
        repeat_threshold = min(repeat_threshold, 16)
    from isaac_aligner_amd import options
    p = options.default_params(*g["read_lengths"], gap_scoring="eland", gapped_mismatches_max=g["gapped_mismatches_max"], semialigned_gap_limit=g["gap_limit"],
                               repeat_threshold=repeat_threshold, min_gap_extend=g["scores"][4])
    assert [p.gap_match, p.gap_mismatch, p.gap_open, p.gap_extend, p.min_gap_extend] == g["scores"]
    p.n_seeds = len(g["seed_offsets"])
    for i, offset in enumerate(g["seed_offsets"]):
        p.seeds[i].offset, p.seeds[i].length, p.seeds[i].read_index = offset, g["seed_length"], i // 3
    return p


def fragment_builder_inputs(case, fixture, seed_id):
    """(bcl [n_clusters, 200] with the test's cluster at its own id, match records, tile) for one test of the suite"""
    cluster = case["matches"][0]["cluster"]
    bcl = np.zeros((cluster + 1, 200), np.uint8)
    bcl[cluster] = np.frombuffer(bytes.fromhex(fixture["clusters"][case["cluster"]]), np.uint8)
    m = np.zeros(len(case["matches"]), abi.MATCH_DTYPE)
    for k, x in enumerate(case["matches"]):
        m["seed_id"][k] = seed_id(x["tile"], 0, x["cluster"], x["seed"], int(x["reverse"]))
        m["location"][k] = ((x["contig"] + 1) << 41) | (x["position"] << 1)
    return bcl, m, case["matches"][0]["tile"]


def check_fragment_builder_case(case, cands, cigars):
    """every value the reference test asserts, on candidates in (read, list order) with CIGAR offsets into `cigars`"""
    exp = case["expected"]
    lists = [cands[cands["read_index"] == r] for r in (0, 1)]
    for r, n in exp["list_sizes"].items():
        assert len(lists[int(r)]) == n, (case["name"], r, len(lists[int(r)]), n)
    if exp["cigar_buffer_size"] is not None:
        assert len(cigars) == exp["cigar_buffer_size"], (case["name"], len(cigars))
    for k, w in exp["cigar_words"].items():
        assert int(cigars[int(k)]) == w, (case["name"], k, int(cigars[int(k)]), w)
    for key, fields in exp["fragments"].items():
        r, j = [int(x) for x in key.split(",")]
        f = lists[r][j]
        for name, v in fields.items():
            if name == "logProbability":
                assert abs(float(f["log_probability"]) - v[0]) <= v[1], (case["name"], key, float(f["log_probability"]), v)
            else:
                assert int(f[CANDIDATE_FIELD[name]]) == v, (case["name"], key, name, int(f[CANDIDATE_FIELD[name]]), v)


def count_record_diffs(a, acig, b, bcig, limit=5, ignore=()):
    """vectorised compare_records for full-size batches: (number of differing records, first few as text)"""
    n = min(len(a), len(b))
    names = [f for f in a.dtype.names if f not in ("cigar_offset", "reserved") + tuple(ignore)]
    eq = np.ones(n, bool)
    for f in names:
        eq &= a[f][:n] == b[f][:n]
    eq &= (a["reserved"][:n] >> 16) == (b["reserved"][:n] >> 16)          # the template's alignment score (bits 16-31; the low bits are diagnostics)
    # CIGARs: word k of every record side by side, for as many words as the longest one has
    la = a["cigar_length"][:n].astype(np.int64)
    oa, ob = a["cigar_offset"][:n].astype(np.int64), b["cigar_offset"][:n].astype(np.int64)
    for k in range(int(la.max()) if n else 0):
        has = eq & (la > k)
        idx = np.nonzero(has)[0]
        if not len(idx):
            break
        eq[idx] &= acig[oa[idx] + k] == bcig[ob[idx] + k]
    bad = np.nonzero(~eq)[0]
    text = []
    for i in bad[:limit]:
        x, y = a[i], b[i]
        text.append("record %d:\n  %s %s\n  %s %s" % (i, x, abi.cigar_string(acig[x["cigar_offset"]:x["cigar_offset"] + x["cigar_length"]]),
                                                      y, abi.cigar_string(bcig[y["cigar_offset"]:y["cigar_offset"] + y["cigar_length"]])))
    return int(len(bad)) + abs(len(a) - len(b)), text
