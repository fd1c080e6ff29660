"""The part of the isaac-align command line that parameterises the hot path: option defaults
(lib/options/AlignOptions.cpp:77-160), the --gap-scoring presets (:55-56,689-743) and the `--seeds auto` rule
(lib/options/alignOptions/SeedDescriptorOption.cpp:90-151; --first-pass-seeds becomes 2 with a non-zero
--semialigned-gap-limit, AlignOptions.cpp:1165-1171)."""
from .abi import MAX_SEEDS, Params

GAP_SCORING = {"bwa": (0, -3, -11, -4, -20), "eland": (2, -1, -15, -3, -25)}


def auto_seeds(read_lengths, seed_length=32):
    """returns ([(offset, read_index)], first_pass_seeds upper bound)"""
    seeds, first_pass = [], 2
    for read_index, length in enumerate(read_lengths):
        ret, generated, offset, end_offset = 1, 0, 0, length
        if length > seed_length:
            seeds.append((0, read_index))
            offset = seed_length
            end_offset = length - seed_length
            seeds.append((end_offset, read_index))
            generated, ret = 2, 2
        while offset + seed_length <= end_offset:
            seeds.append((offset, read_index))
            generated += 1
            offset += seed_length
        offset = seed_length // 2
        if end_offset > seed_length // 2:
            end_offset -= seed_length // 2
            while generated < 4 and offset + seed_length <= end_offset:
                seeds.append((offset, read_index))
                generated += 1
                offset += seed_length
        first_pass = min(first_pass, ret)
    return seeds, first_pass


def default_params(read_length1, read_length2=0, gap_scoring="bwa", **overrides):
    """isaac_params with the reference's defaults for paired (or single) reads of the given lengths"""
    p = Params()
    p.gap_match, p.gap_mismatch, p.gap_open, p.gap_extend, p.min_gap_extend = GAP_SCORING[gap_scoring]
    p.repeat_threshold = 10
    p.gapped_mismatches_max = 5
    p.semialigned_gap_limit = 100
    p.base_quality_cutoff = 25
    p.ignore_neighbors = 0
    p.clip_semialigned = 1
    p.clip_overlapping = 1
    p.scatter_repeats = 0
    p.dodgy_alignment_score = 0
    p.mapq_threshold = 0
    p.keep_unaligned = 1
    p.mate_drift_range = -1
    p.seed_length = 32
    lengths = [read_length1] + ([read_length2] if read_length2 else [])
    p.n_reads = len(lengths)
    for i, length in enumerate(lengths):
        p.read_length[i] = length
    for k, v in overrides.items():
        setattr(p, k, v)
    seeds, first_pass = auto_seeds(lengths, p.seed_length)
    if len(seeds) > MAX_SEEDS:
        raise ValueError("too many seeds")
    p.n_seeds = len(seeds)
    for i, (offset, read_index) in enumerate(seeds):
        p.seeds[i].offset, p.seeds[i].length, p.seeds[i].read_index = offset, p.seed_length, read_index
    if "first_pass_seeds" not in overrides:
        p.first_pass_seeds = min(2 if p.semialigned_gap_limit else 1, first_pass)
    return p


# lib/flowcell/SequencingAdapterMetadata.cpp:29-39 as (sequence, reverse, clip_length; 0 = unbounded): the macros of --default-adapters
ADAPTER_PRESETS = {"Standard": [("AGATCGGAAGAGC", False, 0), ("GCTCTTCCGATCT", True, 0)],
                   "Nextera": [("CTGTCTCTTATACACATCT", False, 0), ("AGATGTGTATAAGAGACAG", True, 0)],
                   "NexteraMp": [("CTGTCTCTTATACACATCT", False, 19), ("AGATGTGTATAAGAGACAG", False, 19)]}


def set_adapters(p, adapters):
    """isaac_params::adapters from a preset name or a list of (sequence, reverse, clip_length) / dicts with those keys; returns p.
    (isaac_gpu_parse_adapters of the library parses the full --default-adapters syntax.)"""
    if isinstance(adapters, str):
        adapters = ADAPTER_PRESETS[adapters]
    adapters = [(a["sequence"], a["reverse"], a["clip_length"]) if isinstance(a, dict) else a for a in adapters]
    if len(adapters) > len(p.adapters):
        raise ValueError("too many adapters")
    p.n_adapters = len(adapters)
    for i, (sequence, reverse, clip_length) in enumerate(adapters):
        p.adapters[i].sequence, p.adapters[i].reverse, p.adapters[i].clip_length = sequence.encode(), int(reverse), int(clip_length)
    return p
