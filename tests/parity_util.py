"""helpers shared by the parity tests: synthetic inputs and record comparison"""
import numpy as np

from isaac_aligner_amd import abi, synth


def make_inputs(genome_bases=300000, n_pairs=2000, read_length=150, read_length2=None, seed=1, n_contigs=2, **kw):
    contigs = synth.make_genome(genome_bases, seed=seed, n_contigs=n_contigs)
    bcl, truth = synth.make_read_pairs(contigs, n_pairs, read_length, seed=seed + 1, read_length2=read_length2, **kw)
    return [bytes(c.numpy()) for c in contigs], bcl.numpy(), truth


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
