"""Host-side mirror of the BAM output boundary (include/isaac_gpu.h: isaac_gpu_bam_records / isaac_gpu_bam_header /
isaac_gpu_bgzf_compress, isaac_gpu_bam_index): what build::Build writes
(reference: lib/build/Build.cpp, include/bam/Bam.hh, include/bgzf/BgzfCompressor.hh)."""
import ctypes as C
import os

import numpy as np


class BamTile(C.Structure):
    _fields_ = [("bcl_dev", C.c_void_p), ("fragments_dev", C.c_void_p), ("cigar_dev", C.c_void_p), ("n_records", C.c_uint64), ("read_name_prefix", C.c_char_p), ("read_group", C.c_char_p), ("tls", C.c_void_p)]


class BamOptions(C.Structure):
    _fields_ = [("forced_dodgy_alignment_score", C.c_uint32), ("pessimistic_mapq", C.c_uint32), ("read_group", C.c_char_p), ("barcode", C.c_char_p),
                ("mark_duplicates", C.c_uint32), ("keep_duplicates", C.c_uint32), ("realign_gaps", C.c_uint32), ("realign_vigorously", C.c_uint32), ("realign_dodgy", C.c_uint32),
                ("tls", C.c_void_p), ("bin_filter", C.c_uint32), ("bin_first_contig", C.c_uint32), ("bin_end_contig", C.c_uint32), ("bin_unaligned", C.c_uint32), ("index_entries_dev", C.c_void_p),
                ("bin_first_position", C.c_uint64), ("bin_end_position", C.c_uint64)]


class BinSize(C.Structure):
    _fields_ = [("n_clusters", C.c_uint64), ("n_cigar_words", C.c_uint64)]


class BinMap(C.Structure):
    _fields_ = [("bin_of_contig", C.c_void_p), ("n_contigs", C.c_uint32), ("cut_positions", C.c_void_p), ("n_cuts", C.c_uint32), ("n_bins", C.c_uint32)]


def reference_position(contig, position):
    """reference::ReferencePosition::getValue() (include/reference/ReferencePosition.hh:51-188), neighbour bit clear"""
    return (((contig + 1) << 40) | position) << 1


class BamError(RuntimeError):
    pass


def _lib():
    from . import gpu
    lib = gpu.load_library()
    lib.isaac_gpu_bam_last_error.restype = C.c_char_p
    lib.isaac_gpu_bgzf_bound.restype = C.c_uint64
    lib.isaac_gpu_bgzf_bound.argtypes = [C.c_uint64]
    return lib


def _check(lib, rc):
    if rc:
        raise BamError("%d: %s" % (rc, lib.isaac_gpu_bam_last_error().decode()))


def _sq_tags(contigs):
    """the optional (AS, UR, M5) of every contig as three char* arrays; contigs: (name, length[, as, ur, m5])"""
    arrays = []
    for k in (2, 3, 4):
        arrays.append((C.c_char_p * max(1, len(contigs)))(*[(c[k].encode() if len(c) > k and c[k] else None) for c in contigs]))
    return arrays


def header(command_line, version, contigs, description="", header_lines=()):
    """bam::serializeHeader; contigs: [(name, length[, AS, UR, M5])] in reference order; header_lines: --bam-header-tag lines and @RG lines"""
    lib = _lib()
    sq_as, sq_ur, sq_m5 = _sq_tags(contigs)
    contigs = [(c[0], c[1]) for c in contigs]
    lines = (C.c_char_p * max(1, len(header_lines)))(*[l.encode() for l in header_lines])
    names = (C.c_char_p * max(1, len(contigs)))(*[n.encode() for n, _ in contigs])
    lengths = (C.c_uint32 * max(1, len(contigs)))(*[l for _, l in contigs])
    n = C.c_uint64()
    capacity = 4096 + len(command_line) + len(description) + sum(len(l) + 1 for l in header_lines) + sum(2 * len(nm) + 64 + 1400 for nm, _ in contigs)
    out = np.empty(capacity, np.uint8)
    _check(lib, lib.isaac_gpu_bam_header(command_line.encode(), description.encode(), version.encode(), lines, C.c_uint32(len(header_lines)), names, lengths,
                                         sq_as, sq_ur, sq_m5, C.c_uint32(len(contigs)), out.ctypes.data_as(C.c_void_p), C.c_uint64(capacity), C.byref(n)))
    return out[:n.value].tobytes()


def bgzf_compress(data, level=1, n_threads=None, eof_block=False):
    """bgzf::BgzfCompressor framing of `data` (bytes or a uint8 numpy array); returns bytes"""
    lib = _lib()
    a = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data, np.uint8)
    capacity = lib.isaac_gpu_bgzf_bound(C.c_uint64(a.size))
    out = np.empty(capacity, np.uint8)
    n = C.c_uint64()
    _check(lib, lib.isaac_gpu_bgzf_compress(a.ctypes.data_as(C.c_void_p), C.c_uint64(a.size), C.c_int(level), C.c_uint32(n_threads or os.cpu_count() or 1), C.c_int(int(eof_block)),
                                            out.ctypes.data_as(C.c_void_p), C.c_uint64(capacity), C.byref(n)))
    return out[:n.value].tobytes()


def write_bam(path, header_bytes, record_bytes, level=1, n_threads=None):
    """header, records and the empty end-of-file block, each flushed as the reference's filter chain flushes them"""
    with open(path, "wb") as f:
        f.write(bgzf_compress(header_bytes, level, n_threads))
        f.write(bgzf_compress(record_bytes, level, n_threads, eof_block=True))


class IndexPart(C.Structure):
    _fields_ = [("records_offset", C.c_uint64), ("records_bytes", C.c_uint64), ("bgzf_host", C.c_void_p), ("bgzf_bytes", C.c_uint64)]


def split_parts(record_bytes, unaligned_offset):
    """the bins of the file as isaac-align makes them: [(offset, bytes)] of every contig's records, then of the unaligned ones"""
    data = memoryview(record_bytes)
    parts, at = [], 0
    while at < unaligned_offset:
        contig, begin = bytes(data[at + 4:at + 8]), at
        while at < unaligned_offset and bytes(data[at + 4:at + 8]) == contig:
            at += 4 + int.from_bytes(data[at:at + 4], "little")
        parts.append((begin, at - begin))
    if unaligned_offset < len(data):
        parts.append((unaligned_offset, len(data) - unaligned_offset))
    return parts


def index(record_bytes, parts, n_contigs, header_bgzf_bytes):
    """sorted.bam.bai (bam::BamIndex): parts = [(records_offset, records_bytes, bgzf bytes)] in file order; returns the file's bytes"""
    lib = _lib()
    lib.isaac_gpu_bam_index_last_error.restype = C.c_char_p
    records = np.frombuffer(record_bytes, np.uint8) if isinstance(record_bytes, (bytes, bytearray, memoryview)) else np.ascontiguousarray(record_bytes, np.uint8)
    arr = (IndexPart * max(1, len(parts)))()
    keep = []
    for i, (offset, n, bgzf) in enumerate(parts):
        keep.append(np.frombuffer(bgzf, np.uint8))
        arr[i].records_offset, arr[i].records_bytes, arr[i].bgzf_host, arr[i].bgzf_bytes = offset, n, keep[-1].ctypes.data, len(bgzf)
    n = C.c_uint64()
    lib.isaac_gpu_bam_index(records.ctypes.data_as(C.c_void_p), arr, C.c_uint32(len(parts)), C.c_uint32(n_contigs), C.c_uint64(header_bgzf_bytes), None, C.c_uint64(0), C.byref(n))
    out = np.empty(max(1, n.value), np.uint8)
    rc = lib.isaac_gpu_bam_index(records.ctypes.data_as(C.c_void_p), arr, C.c_uint32(len(parts)), C.c_uint32(n_contigs), C.c_uint64(header_bgzf_bytes), out.ctypes.data_as(C.c_void_p),
                                 C.c_uint64(out.size), C.byref(n))
    if rc:
        raise BamError("%d: %s" % (rc, lib.isaac_gpu_bam_index_last_error().decode()))
    return out[:n.value].tobytes()


def parse_index(bai):
    """decodes a .bai file: (per contig: {"bins": {bin: [(begin, end)]}, "stats": (begin, end, mapped, unmapped) or None, "linear": [offsets]}, n_no_coordinate)"""
    assert bai[:4] == b"BAI\1"
    n_ref, at = int.from_bytes(bai[4:8], "little"), 8
    contigs = []
    for _ in range(n_ref):
        n_bin = int.from_bytes(bai[at:at + 4], "little"); at += 4
        bins, stats = {}, None
        for _ in range(n_bin):
            b, n_chunk = int.from_bytes(bai[at:at + 4], "little"), int.from_bytes(bai[at + 4:at + 8], "little"); at += 8
            chunks = [(int.from_bytes(bai[at + 16 * k:at + 16 * k + 8], "little"), int.from_bytes(bai[at + 16 * k + 8:at + 16 * k + 16], "little")) for k in range(n_chunk)]
            at += 16 * n_chunk
            if b == 37450:
                stats = (chunks[0][0], chunks[0][1], chunks[1][0], chunks[1][1])
            else:
                bins[b] = chunks
        n_intv = int.from_bytes(bai[at:at + 4], "little"); at += 4
        linear = [int.from_bytes(bai[at + 8 * k:at + 8 * k + 8], "little") for k in range(n_intv)]; at += 8 * n_intv
        contigs.append(dict(bins=bins, stats=stats, linear=linear))
    no_coordinate = int.from_bytes(bai[at:at + 8], "little"); at += 8
    assert at == len(bai)
    return contigs, no_coordinate


def parse_records(data):
    """decodes an uncompressed BAM record stream into dicts (for tests and examples)"""
    out, at = [], 0
    data = bytes(data)
    while at < len(data):
        size = int.from_bytes(data[at:at + 4], "little")
        b = data[at + 4:at + 4 + size]
        ref_id, pos, bin_mq_nl, flag_nc, l_seq, next_ref, next_pos, tlen = np.frombuffer(b[:32], "<i4")
        nl, mq, bn = bin_mq_nl & 0xff, (bin_mq_nl >> 8) & 0xff, (int(bin_mq_nl) >> 16) & 0xffff
        nc, flag = flag_nc & 0xffff, (int(flag_nc) >> 16) & 0xffff
        p = 32
        name = b[p:p + nl - 1].decode(); p += nl
        cigar = np.frombuffer(b[p:p + 4 * nc], "<u4"); p += 4 * nc
        seq4 = np.frombuffer(b[p:p + (l_seq + 1) // 2], np.uint8); p += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(seq4[i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = np.frombuffer(b[p:p + l_seq], np.uint8); p += l_seq
        tags = {}
        while p < len(b):
            tag, typ = b[p:p + 2].decode(), chr(b[p + 2]); p += 3
            if typ == "i":
                tags[tag] = int.from_bytes(b[p:p + 4], "little", signed=True); p += 4
            elif typ == "Z":
                e = b.index(0, p); tags[tag] = b[p:e].decode(); p = e + 1
            else:
                raise ValueError("tag type " + typ)
        out.append(dict(ref_id=int(ref_id), pos=int(pos), bin=bn, mapq=int(mq), flag=flag, name=name, cigar=cigar, seq=seq, qual=qual, next_ref_id=int(next_ref), next_pos=int(next_pos),
                        tlen=int(tlen), tags=tags))
        at += 4 + size
    return out
