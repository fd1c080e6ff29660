"""ctypes / numpy mirrors of include/isaac_gpu.h (record layouts of the C ABI)."""
import ctypes as C

import numpy as np

MAX_SEEDS = 16
MAX_CIGAR_OPS = 40
MAX_ADAPTERS = 8


class Seed(C.Structure):
    _fields_ = [("offset", C.c_uint16), ("length", C.c_uint16), ("read_index", C.c_uint32)]


class Adapter(C.Structure):
    """isaac_adapter"""
    _fields_ = [("sequence", C.c_char * 128), ("reverse", C.c_uint32), ("clip_length", C.c_uint32)]


class Params(C.Structure):
    """isaac_params"""
    _fields_ = [("gap_match", C.c_int32), ("gap_mismatch", C.c_int32), ("gap_open", C.c_int32), ("gap_extend", C.c_int32), ("min_gap_extend", C.c_int32),
                ("repeat_threshold", C.c_uint32), ("gapped_mismatches_max", C.c_uint32), ("semialigned_gap_limit", C.c_uint32), ("base_quality_cutoff", C.c_uint32),
                ("ignore_neighbors", C.c_uint32), ("clip_semialigned", C.c_uint32), ("clip_overlapping", C.c_uint32), ("scatter_repeats", C.c_uint32),
                ("dodgy_alignment_score", C.c_int32), ("mapq_threshold", C.c_uint32), ("keep_unaligned", C.c_uint32), ("mate_drift_range", C.c_int32),
                ("first_pass_seeds", C.c_uint32), ("seed_length", C.c_uint32),
                ("n_reads", C.c_uint32), ("read_length", C.c_uint32 * 2), ("n_seeds", C.c_uint32), ("seeds", Seed * MAX_SEEDS),
                ("n_adapters", C.c_uint32), ("adapters", Adapter * MAX_ADAPTERS)]


class Tls(C.Structure):
    """isaac_tls"""
    _fields_ = [("min", C.c_uint32), ("max", C.c_uint32), ("median", C.c_uint32), ("low_std_dev", C.c_uint32), ("high_std_dev", C.c_uint32),
                ("best_model", C.c_int32 * 2), ("stable", C.c_uint32), ("mate_min", C.c_uint32), ("mate_max", C.c_uint32)]

    def astuple(self):
        return (self.min, self.max, self.median, self.low_std_dev, self.high_std_dev, self.best_model[0], self.best_model[1], self.stable, self.mate_min, self.mate_max)


class Counters(C.Structure):
    """isaac_counters"""
    _fields_ = [(n, C.c_uint64) for n in ("clusters", "probes", "probe_steps", "matches", "candidates", "ungapped_scans", "bsw_jobs", "bsw_accepted", "simple_indels",
                                          "rescue_calls", "rescue_window_bases", "rescue_candidates", "rescue_bsw", "overflow_clusters", "mapq_near_integer", "heavy_clusters",
                                          "residual_capacity", "residual_near_tie", "residual_oversize", "large_sums")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


MATCH_DTYPE = np.dtype([("seed_id", "<u8"), ("location", "<u8")])
REFERENCE_KMER_DTYPE = np.dtype([("kmer", "<u8"), ("position", "<u8")])
CANDIDATE_DTYPE = np.dtype([("position", "<i8"), ("log_probability", "<f8"), ("cluster", "<u4"), ("read_index", "<u4"), ("contig_id", "<u4"),
                            ("observed_length", "<u4"), ("reverse", "<u4"), ("mismatch_count", "<u4"), ("matches_in_a_row", "<u4"), ("gap_count", "<u4"),
                            ("edit_distance", "<u4"), ("smith_waterman_score", "<u4"), ("unique_seed_count", "<u4"), ("non_unique_first", "<u4"),
                            ("non_unique_second", "<u4"), ("repeat_seeds_count", "<u4"), ("cigar_offset", "<u4"), ("cigar_length", "<u4"),
                            ("low_clipped", "<u4"), ("high_clipped", "<u4"), ("first_seed_index", "<i4"), ("reserved", "<u4")])
FRAGMENT_DTYPE = np.dtype([("f_strand_position", "<u8"), ("mate_f_strand_position", "<u8"), ("bam_tlen", "<i4"), ("observed_length", "<u4"),
                           ("low_clipped", "<u2"), ("high_clipped", "<u2"), ("alignment_score", "<u2"), ("template_alignment_score", "<u2"),
                           ("read_length", "<u2"), ("cigar_length", "<u2"), ("gap_count", "<u2"), ("edit_distance", "<u2"),
                           ("flags", "<u4"), ("cigar_offset", "<u4"), ("tile", "<u4"), ("cluster_id", "<u4"), ("mapq", "<u4"), ("reserved", "<u4")])
BSW_JOB_DTYPE = np.dtype([("query_offset", "<u8"), ("database_offset", "<u8"), ("query_length", "<u4"), ("reserved", "<u4")])
BSW_RESULT_DTYPE = np.dtype([("n_ops", "<u4"), ("offset", "<u4"), ("cigar", "<u4", (MAX_CIGAR_OPS,))])
assert CANDIDATE_DTYPE.itemsize == 96 and FRAGMENT_DTYPE.itemsize == 64 and BSW_RESULT_DTYPE.itemsize == 8 + 4 * MAX_CIGAR_OPS

REFPOS_NOMATCH = ((2 ** 64 - 1) >> 41) << 41
CIGAR_OPS = "MIDNSHP=X?"


def cigar_string(words):
    return "".join("%d%s" % (int(w) >> 4, CIGAR_OPS[min(int(w) & 0xF, 9)]) for w in words)


def refpos_contig(v):
    return (np.asarray(v, np.uint64) >> np.uint64(41)).astype(np.int64) - 1


def refpos_position(v):
    return ((np.asarray(v, np.uint64) >> np.uint64(1)) & np.uint64((1 << 40) - 1)).astype(np.int64)
