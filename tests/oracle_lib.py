"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product package
(isaac_aligner_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class Seed(C.Structure):
    _fields_ = [("offset", C.c_uint16), ("length", C.c_uint16), ("read_index", C.c_uint32)]


class Adapter(C.Structure):
    _fields_ = [("sequence", C.c_char * 128), ("reverse", C.c_uint32), ("clip_length", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("gap_match", C.c_int32), ("gap_mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("min_gap_extend", C.c_int32),
                ("repeat_threshold", C.c_uint32), ("gapped_mismatches_max", C.c_uint32), ("semialigned_gap_limit", C.c_uint32), ("base_quality_cutoff", C.c_uint32),
                ("ignore_neighbors", C.c_uint32), ("clip_semialigned", C.c_uint32), ("clip_overlapping", C.c_uint32), ("scatter_repeats", C.c_uint32),
                ("dodgy_alignment_score", C.c_int32), ("mapq_threshold", C.c_uint32), ("keep_unaligned", C.c_uint32), ("mate_drift_range", C.c_int32),
                ("first_pass_seeds", C.c_uint32), ("seed_length", C.c_uint32),
                ("n_reads", C.c_uint32), ("read_length", C.c_uint32 * 2), ("n_seeds", C.c_uint32), ("seeds", Seed * 16),
                ("n_adapters", C.c_uint32), ("adapters", Adapter * 8)]


class Tls(C.Structure):
    _fields_ = [("min", C.c_uint32), ("max", C.c_uint32), ("median", C.c_uint32), ("low_std_dev", C.c_uint32), ("high_std_dev", C.c_uint32),
                ("best_model", C.c_int32 * 2), ("stable", C.c_uint32), ("mate_min", C.c_uint32), ("mate_max", C.c_uint32)]

    def astuple(self):
        return (self.min, self.max, self.median, self.low_std_dev, self.high_std_dev, self.best_model[0], self.best_model[1], self.stable, self.mate_min, self.mate_max)


CANDIDATE_DTYPE = np.dtype([("position", "<i8"), ("log_probability", "<f8"), ("cluster", "<u4"), ("read_index", "<u4"), ("contig_id", "<u4"),
                            ("observed_length", "<u4"), ("reverse", "<u4"), ("mismatch_count", "<u4"), ("matches_in_a_row", "<u4"), ("gap_count", "<u4"),
                            ("edit_distance", "<u4"), ("smith_waterman_score", "<u4"), ("unique_seed_count", "<u4"), ("non_unique_first", "<u4"),
                            ("non_unique_second", "<u4"), ("repeat_seeds_count", "<u4"), ("cigar_offset", "<u4"), ("cigar_length", "<u4"),
                            ("low_clipped", "<u4"), ("high_clipped", "<u4"), ("first_seed_index", "<i4"), ("reserved", "<u4")])
assert CANDIDATE_DTYPE.itemsize == 96

RECORD_DTYPE = np.dtype([("f_strand_position", "<u8"), ("mate_f_strand_position", "<u8"), ("bam_tlen", "<i4"), ("observed_length", "<u4"),
                         ("low_clipped", "<u2"), ("high_clipped", "<u2"), ("alignment_score", "<u2"), ("template_alignment_score", "<u2"),
                         ("read_length", "<u2"), ("cigar_length", "<u2"), ("gap_count", "<u2"), ("edit_distance", "<u2"),
                         ("flags", "<u4"), ("cigar_offset", "<u4"), ("tile", "<u4"), ("cluster_id", "<u4"), ("mapq", "<u4"), ("reserved", "<u4")])
assert RECORD_DTYPE.itemsize == 64

MATCH_DTYPE = np.dtype([("seed_id", "<u8"), ("location", "<u8")])
INDEX_DTYPE = np.dtype([("kmer", "<u8"), ("position", "<u8")])

CIGAR_OPS = "MIDNSHP=X?"


class BamTile(C.Structure):
    _fields_ = [("bcl", C.c_void_p), ("records", C.c_void_p), ("cigars", C.c_void_p), ("n_records", C.c_uint64), ("read_name_prefix", C.c_char_p), ("read_group", C.c_char_p), ("tls", C.c_void_p)]


def cigar_string(words):
    return "".join("%d%s" % (int(w) >> 4, CIGAR_OPS[min(int(w) & 0xF, 9)]) for w in words)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_last_error.restype = C.c_char_p
        lib.oracle_ref_create.restype = C.c_void_p
        lib.oracle_ref_index_size.restype = C.c_uint64

    def check(self, rc):
        if rc:
            raise RuntimeError(self.lib.oracle_last_error().decode())

    def realign_case(self, case, realigner):
        """one testGapRealigner case (tests/golden/gap_realigner.json) through GapRealigner::realign; returns a dict of what the test looks at"""
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        bcl = np.array([(code[b] | 0x20) if b != "N" else 0 for b in case["read_bases"]], np.uint8)      # TestFragmentAccessor: quality 8, N = 0
        contig = case["contig"].encode()
        cigar = np.array(case["cigar"], np.uint32)
        gp = np.array([g[0] for g in case["gaps"]], np.int64); gl = np.array([g[1] for g in case["gaps"]], np.int32)
        pos, ncig, ed, obs, nov = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        out_cigar, overlaps = np.zeros(4096, np.uint32), np.zeros(32, np.uint32)
        self.check(self.lib.oracle_realign_case(contig, C.c_uint64(len(contig)), ptr(bcl), C.c_uint32(len(bcl)), C.c_uint64(case["f_strand_position"]), ptr(cigar), C.c_uint32(len(cigar)),
                                                C.c_uint32(case["observed_length"]), C.c_uint32(case["edit_distance"]), C.c_uint32(case["low_clipped"]), C.c_uint32(case["high_clipped"]),
                                                ptr(gp), ptr(gl), C.c_uint32(len(gp)), C.c_uint32(case["mismatch_cost"]), C.c_uint32(case["gap_open_cost"]), C.c_uint32(realigner["gap_extend_cost"]),
                                                C.c_int(int(realigner["vigorous"])), C.c_int(int(realigner["dodgy"])), C.c_uint32(realigner["gaps_per_fragment"]), C.c_int(int(realigner["clip_semialigned"])),
                                                C.c_uint64(case["bin_start"]), C.c_int64(-1 if case["bin_end"] is None else case["bin_end"]),
                                                C.byref(pos), ptr(out_cigar), C.byref(ncig), C.byref(ed), C.byref(obs), ptr(overlaps), C.byref(nov)))
        return {"position": pos.value, "cigar": cigar_string(out_cigar[:ncig.value]), "edit_distance": ed.value, "observed_length": obs.value, "overlaps": [int(x) for x in overlaps[:nov.value]]}

    def filter_duplicates(self, primary, mate_anchor, mate_info, rank, cluster_id):
        """DuplicatePairEndFilter over literal index entries; returns a bool array: entry i is a duplicate of a better one"""
        arrays = [np.ascontiguousarray(primary, np.uint64), np.ascontiguousarray(mate_anchor, np.uint64), np.ascontiguousarray(mate_info, np.uint32),
                  np.ascontiguousarray(rank, np.uint64), np.ascontiguousarray(cluster_id, np.uint64)]
        out = np.zeros(len(arrays[0]), np.uint8)
        self.check(self.lib.oracle_filter_duplicates(C.c_uint64(len(out)), *[ptr(a) for a in arrays], ptr(out)))
        return out.astype(bool)

    def bam_records(self, tiles, read_lengths, forced_dodgy_alignment_score=0, pessimistic_mapq=False, read_group="0", barcode="none", mark_duplicates=False, keep_duplicates=True,
                    realign_gaps=False, realign_dodgy=False, clip_semialigned=True, reference=None, tls=None, bin_cuts=(), realign_vigorously=False):
        """tiles: [(bcl, records, cigars, read_name_prefix[, read_group[, tls]])] as numpy arrays; bin_cuts: ascending ReferencePosition values at which a
        contig goes on into a further bin (every bin is filtered and realigned by itself); returns (bytes, n_records, unaligned_offset)"""
        arr = (BamTile * len(tiles))()
        keep = []
        total = 0
        for i, tile in enumerate(tiles):
            bcl, records, cigars, prefix = tile[:4]
            bcl, records, cigars = np.ascontiguousarray(bcl, np.uint8), np.ascontiguousarray(records), np.ascontiguousarray(cigars, np.uint32)
            keep.append((bcl, records, cigars, prefix.encode()))
            arr[i].bcl = bcl.ctypes.data; arr[i].records = records.ctypes.data; arr[i].cigars = cigars.ctypes.data
            arr[i].n_records = len(records); arr[i].read_name_prefix = keep[-1][3]
            if len(tile) > 4 and tile[4] is not None:
                keep.append(tile[4].encode()); arr[i].read_group = keep[-1]
            if len(tile) > 5 and tile[5] is not None:
                arr[i].tls = C.cast(C.pointer(tile[5]), C.c_void_p)
            total += len(records)
        lengths = (C.c_uint32 * 2)(*(list(read_lengths) + [0])[:2])
        cap = max(1, total * (128 + 2 * max(read_lengths) + 4 * 64))
        out = np.empty(cap, np.uint8)
        nb, nr, un = C.c_uint64(), C.c_uint64(), C.c_uint64()
        cuts = np.ascontiguousarray(list(bin_cuts), np.uint64)
        self.check(self.lib.oracle_bam_records_cuts(arr, C.c_uint32(len(tiles)), C.c_uint32(len(read_lengths)), lengths, C.c_uint32(forced_dodgy_alignment_score), C.c_int(int(pessimistic_mapq)),
                                                    read_group.encode(), barcode.encode(), C.c_int(int(mark_duplicates)), C.c_int(int(keep_duplicates)),
                                                    C.c_int(2 if (realign_gaps and realign_vigorously) else int(realign_gaps)), C.c_int(int(realign_dodgy)), C.c_int(int(clip_semialigned)), (reference.h if reference is not None else None),
                                                    C.byref(tls) if tls is not None else None, ptr(cuts) if len(cuts) else None, C.c_uint32(len(cuts)),
                                                    ptr(out), C.c_uint64(cap), C.byref(nb), C.byref(nr), C.byref(un)))
        return out[:nb.value].tobytes(), nr.value, un.value

    def bam_index(self, record_bytes, parts, n_contigs, header_bgzf_bytes):
        """parts: [(records_offset, records_bytes, bgzf bytes)] in file order; returns the .bai bytes"""
        records = np.frombuffer(record_bytes, np.uint8)
        keep = [np.frombuffer(p[2], np.uint8) for p in parts]
        offsets = (C.c_uint64 * max(1, len(parts)))(*[p[0] for p in parts])
        sizes = (C.c_uint64 * max(1, len(parts)))(*[p[1] for p in parts])
        bgzf = (C.c_void_p * max(1, len(parts)))(*[k.ctypes.data for k in keep])
        bgzf_sizes = (C.c_uint64 * max(1, len(parts)))(*[len(p[2]) for p in parts])
        out = np.empty(64 + 16 * len(records) // 32 + 400000 * n_contigs, np.uint8)
        n = C.c_uint64()
        self.check(self.lib.oracle_bam_index(ptr(records), offsets, sizes, bgzf, bgzf_sizes, C.c_uint32(len(parts)), C.c_uint32(n_contigs), C.c_uint32(header_bgzf_bytes),
                                             ptr(out), C.c_uint64(out.size), C.byref(n)))
        return out[:n.value].tobytes()

    def bam_header(self, command_line, version, contigs, description="", header_lines=()):
        lines = (C.c_char_p * max(1, len(header_lines)))(*[l.encode() for l in header_lines])
        names = (C.c_char_p * max(1, len(contigs)))(*[c[0].encode() for c in contigs])
        lengths = (C.c_uint32 * max(1, len(contigs)))(*[c[1] for c in contigs])
        tags = [(C.c_char_p * max(1, len(contigs)))(*[(c[k].encode() if len(c) > k and c[k] else None) for c in contigs]) for k in (2, 3, 4)]
        out = np.empty(1 << 20, np.uint8)
        n = C.c_uint64()
        self.check(self.lib.oracle_bam_header(command_line.encode(), description.encode(), version.encode(), lines, C.c_uint32(len(header_lines)), names, lengths, tags[0], tags[1], tags[2],
                                              C.c_uint32(len(contigs)),
                                              ptr(out), C.c_uint64(out.size), C.byref(n)))
        return out[:n.value].tobytes()

    def fastq_tiles(self, clusters_loaded, n_seeds, clusters_at_a_time=0, first_tile=1):
        numbers, sizes = (C.c_uint32 * 4096)(), (C.c_uint32 * 4096)()
        n, nxt = C.c_uint32(), C.c_uint32()
        self.check(self.lib.oracle_fastq_tiles(C.c_uint32(clusters_loaded), C.c_uint32(clusters_at_a_time), C.c_uint32(n_seeds), C.c_uint32(first_tile), numbers, sizes, C.c_uint32(4096),
                                               C.byref(n), C.byref(nxt)))
        return [(numbers[i], sizes[i]) for i in range(n.value)], nxt.value

    def default_params(self, n_reads, len1, len2=0):
        p = Params()
        self.check(self.lib.oracle_default_params(C.c_uint32(n_reads), C.c_uint32(len1), C.c_uint32(len2), C.byref(p)))
        return p

    def bsw_align(self, scores, max_read_length, query, database):
        q, d = query.encode() if isinstance(query, str) else bytes(query), database.encode() if isinstance(database, str) else bytes(database)
        assert len(d) == len(q) + 15
        cig = np.zeros(1024, np.uint32)
        n, off = C.c_uint32(), C.c_uint32()
        self.check(self.lib.oracle_bsw_align(scores[0], scores[1], scores[2], scores[3], max_read_length, q, C.c_uint32(len(q)), d, ptr(cig), C.c_uint32(1024), C.byref(n), C.byref(off)))
        return cig[:n.value].copy(), off.value

    def bsw_force_scalar(self, on):
        """this thread's banded Smith-Waterman rows lane by lane (True) or by the AVX2 form (False)"""
        self.lib.oracle_bsw_force_scalar(C.c_int(int(on)))

    def bsw_check(self, match, mismatch, gap_open, gap_extend, max_read_length):
        return bool(self.lib.oracle_bsw_check(match, mismatch, gap_open, gap_extend, max_read_length))

    def fastq_to_bcl(self, text, read_length, allow_variable_length=False, final=True, max_clusters=None, bcl=None, cluster_stride=None, offset=0):
        """io::FastqReader + FastqLoader::loadSingleRead over a memory buffer.  Returns (rc, bcl[n, stride], n_clusters, consumed, error_offset)"""
        data = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, np.uint8)
        if max_clusters is None:
            max_clusters = len(data) // 4 + 1
        stride = cluster_stride or read_length
        if bcl is None:
            bcl = np.full((max_clusters, stride), 0xEE, np.uint8)
        n, consumed, err = C.c_uint32(), C.c_uint64(), C.c_uint64()
        self.lib.oracle_fastq_to_bcl.restype = C.c_int
        rc = self.lib.oracle_fastq_to_bcl(ptr(data), C.c_uint64(len(data)), C.c_uint32(read_length), int(allow_variable_length), int(final),
                                          C.c_void_p(bcl.ctypes.data + offset), C.c_uint64(stride), C.c_uint32(max_clusters), C.byref(n), C.byref(consumed), C.byref(err))
        return rc, bcl, n.value, consumed.value, err.value

    def seed_id(self, tile, barcode, cluster, seed, reverse):
        v = C.c_uint64()
        self.check(self.lib.oracle_seed_id(C.c_uint64(tile), C.c_uint64(barcode), C.c_uint64(cluster), C.c_uint64(seed), C.c_uint64(reverse), C.byref(v)))
        return v.value

    def simple_indel_literal(self, read, reference, seeds, left_clip0, right_clip1):
        out = np.zeros(2, CANDIDATE_DTYPE)
        cig = np.zeros(256, np.uint32)
        n = C.c_uint64()
        so = (C.c_uint32 * 2)(*(seeds or [0, 0]))
        self.check(self.lib.oracle_simple_indel_literal(read.encode(), reference.encode(), 1 if seeds else 0, so, C.c_uint32(left_clip0), C.c_uint32(right_clip1),
                                                        ptr(out), ptr(cig), C.c_uint64(256), C.byref(n)))
        return out, cig[:n.value].copy()

    def fragment_builder2_literal(self, read, reference, reverse, position, gapped):
        out = np.zeros(1, CANDIDATE_DTYPE)
        cig = np.zeros(256, np.uint32)
        n, cyc = C.c_uint64(), C.c_uint32()
        self.check(self.lib.oracle_fragment_builder2_literal(read.encode(), reference.encode(), int(reverse), int(position is not None), C.c_int64(position or 0), int(gapped),
                                                             ptr(out), ptr(cig), C.c_uint64(256), C.byref(n), C.byref(cyc)))
        return out[0], cig[:n.value].copy(), cyc.value

    def sequencing_adapter_literal(self, read, reference, reverse, adapters):
        """adapters: list of dicts sequence / reverse / clip_length (0: unbounded)"""
        out = np.zeros(1, CANDIDATE_DTYPE)
        cig = np.zeros(256, np.uint32)
        n = C.c_uint64()
        k = len(adapters)
        sequences = (C.c_char_p * k)(*[a["sequence"].encode() for a in adapters])
        rev = (C.c_uint32 * k)(*[int(a["reverse"]) for a in adapters])
        clip = (C.c_uint32 * k)(*[int(a["clip_length"]) for a in adapters])
        self.check(self.lib.oracle_sequencing_adapter_literal(read.encode(), reference.encode(), int(reverse), C.c_uint32(k), sequences, rev, clip, ptr(out), ptr(cig), C.c_uint64(256), C.byref(n)))
        return out[0], cig[:n.value].copy()

    def reference(self, contigs):
        return OracleReference(self, contigs)


class OracleReference:
    """contigs (list of bytes/str, ASCII ACGTN) + sorted 32-mer index"""

    def __init__(self, oracle, contigs):
        self.o = oracle
        self.contigs = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
        self.bases = np.frombuffer(b"".join(self.contigs), np.uint8).copy()
        self.offsets = np.zeros(len(self.contigs) + 1, np.uint64)
        self.offsets[1:] = np.cumsum([len(c) for c in self.contigs])
        self.h = C.c_void_p(oracle.lib.oracle_ref_create(ptr(self.bases), ptr(self.offsets), C.c_uint32(len(self.contigs))))

    def __del__(self):
        try:
            self.o.lib.oracle_ref_destroy(self.h)
        except Exception:
            pass

    def build_index(self, repeat_threshold=1000, annotate_neighbors=True, neighborhood_width=4, n_threads=1):
        """ReferenceSorter + NeighborsFinder restated; n_threads > 1 only changes how fast (sorts and neighbour stretches in parallel)"""
        if n_threads > 1:
            self.o.check(self.o.lib.oracle_ref_build_index_mt(self.h, C.c_uint32(repeat_threshold), int(annotate_neighbors), C.c_uint32(neighborhood_width), C.c_uint32(n_threads)))
            return self.index()
        self.o.check(self.o.lib.oracle_ref_build_index(self.h, C.c_uint32(repeat_threshold), int(annotate_neighbors), C.c_uint32(neighborhood_width)))
        return self.index()

    def index(self):
        n = self.o.lib.oracle_ref_index_size(self.h)
        out = np.zeros(n, INDEX_DTYPE)
        self.o.lib.oracle_ref_get_index(self.h, ptr(out))
        return out

    def set_index(self, index):
        index = np.ascontiguousarray(index, INDEX_DTYPE)
        self.o.lib.oracle_ref_set_index(self.h, ptr(index), C.c_uint64(len(index)))

    def set_karyotype(self, karyotype):
        k = np.ascontiguousarray(karyotype, np.uint32)
        self.o.lib.oracle_ref_set_karyotype(self.h, ptr(k), C.c_uint32(len(k)))

    def find_matches(self, params, bcl, n_clusters, tile=0, n_threads=1):
        cap = int(n_clusters) * 16 * 10 + 1024
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_uint64()
        hits = np.zeros(len(self.contigs), np.uint8)
        if n_threads > 1:
            self.o.check(self.o.lib.oracle_find_matches_mt(self.h, C.byref(params), ptr(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), C.c_uint32(n_threads),
                                                           ptr(out), C.c_uint64(cap), C.byref(n), ptr(hits)))
        else:
            self.o.check(self.o.lib.oracle_find_matches(self.h, C.byref(params), ptr(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), ptr(out), C.c_uint64(cap), C.byref(n), ptr(hits)))
        return out[:n.value].copy(), hits

    def build_fragments(self, params, bcl, matches, contig_loaded=None, tile=0, with_gaps=True, trim=True):
        cap = max(len(matches), 16) + 1024
        out = np.zeros(cap, CANDIDATE_DTYPE)
        cig = np.zeros(cap * 8, np.uint32)
        n, nc = C.c_uint64(), C.c_uint64()
        matches = np.ascontiguousarray(matches, MATCH_DTYPE)
        cl = ptr(np.ascontiguousarray(contig_loaded, np.uint8)) if contig_loaded is not None else None
        self.o.check(self.o.lib.oracle_build_fragments(self.h, C.byref(params), cl, ptr(bcl), C.c_uint32(tile), ptr(matches), C.c_uint64(len(matches)), int(with_gaps), int(trim),
                                                       ptr(out), C.c_uint64(cap), C.byref(n), ptr(cig), C.c_uint64(len(cig)), C.byref(nc)))
        return out[:n.value].copy(), cig[:nc.value].copy()

    def determine_tls(self, params, bcl, matches, contig_loaded=None, tile=0):
        t = Tls()
        matches = np.ascontiguousarray(matches, MATCH_DTYPE)
        cl = ptr(np.ascontiguousarray(contig_loaded, np.uint8)) if contig_loaded is not None else None
        self.o.check(self.o.lib.oracle_determine_tls(self.h, C.byref(params), cl, ptr(bcl), C.c_uint32(tile), ptr(matches), C.c_uint64(len(matches)), C.byref(t)))
        return t

    def select(self, params, bcl, matches, tls, contig_loaded=None, tile=0, n_threads=1, n_clusters_hint=None):
        matches = np.ascontiguousarray(matches, MATCH_DTYPE)
        cap = 2 * (int(n_clusters_hint) if n_clusters_hint else len(matches)) + 64
        out = np.zeros(cap, RECORD_DTYPE)
        cig = np.zeros(cap * 8, np.uint32)
        n, nc = C.c_uint64(), C.c_uint64()
        counters = (C.c_uint64 * 2)()
        cl = ptr(np.ascontiguousarray(contig_loaded, np.uint8)) if contig_loaded is not None else None
        self.o.check(self.o.lib.oracle_select(self.h, C.byref(params), cl, ptr(bcl), C.c_uint32(tile), ptr(matches), C.c_uint64(len(matches)), C.byref(tls), C.c_uint32(n_threads),
                                              ptr(out), C.c_uint64(cap), C.byref(n), ptr(cig), C.c_uint64(len(cig)), C.byref(nc), counters))
        return out[:n.value].copy(), cig[:nc.value].copy(), (counters[0], counters[1])


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-j8"])


_cached = None


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        _cached = Oracle(C.CDLL(so))
    return _cached
