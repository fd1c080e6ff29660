"""sorted-reference.xml (reference::SortedReferenceMetadata, lib/reference/SortedReferenceXml.cpp) through the C ABI:
isaac_gpu_sorted_reference_parse / _format work on memory buffers and need no GPU."""
import ctypes as C

from . import gpu


class Contig(C.Structure):
    """isaac_reference_contig"""
    _fields_ = [("genomic_position", C.c_uint64), ("offset", C.c_uint64), ("size", C.c_uint64), ("total_bases", C.c_uint64), ("acgt_bases", C.c_uint64),
                ("index", C.c_uint32), ("karyotype_index", C.c_uint32),
                ("name", C.c_char * 256), ("file", C.c_char * 1024), ("bam_sq_as", C.c_char * 256), ("bam_sq_ur", C.c_char * 1024), ("bam_m5", C.c_char * 64)]


class MaskFile(C.Structure):
    """isaac_reference_mask_file"""
    _fields_ = [("kmers", C.c_uint64), ("mask_width", C.c_uint32), ("mask", C.c_uint32), ("seed_length", C.c_uint32), ("reserved", C.c_uint32), ("file", C.c_char * 1024)]


class FormatError(ValueError):
    pass


def last_error(lib=None):
    lib = lib or gpu.load_library()
    lib.isaac_gpu_sorted_reference_last_error.restype = C.c_char_p
    return lib.isaac_gpu_sorted_reference_last_error().decode()


def parse(xml_text):
    """returns (contigs, masks, format_version) as lists of Contig / MaskFile"""
    lib = gpu.load_library()
    data = xml_text.encode() if isinstance(xml_text, str) else bytes(xml_text)
    nc, nm, version = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib.isaac_gpu_sorted_reference_parse(data, C.c_uint64(len(data)), None, C.c_uint32(0), C.byref(nc), None, C.c_uint32(0), C.byref(nm), C.byref(version))
    if rc:
        raise FormatError(last_error(lib))
    contigs, masks = (Contig * max(1, nc.value))(), (MaskFile * max(1, nm.value))()
    rc = lib.isaac_gpu_sorted_reference_parse(data, C.c_uint64(len(data)), contigs, nc, C.byref(nc), masks, nm, C.byref(nm), C.byref(version))
    if rc:
        raise FormatError(last_error(lib))
    return list(contigs)[:nc.value], list(masks)[:nm.value], version.value


def format(contigs, masks):
    lib = gpu.load_library()
    ca, ma = (Contig * max(1, len(contigs)))(*contigs), (MaskFile * max(1, len(masks)))(*masks)
    n = C.c_uint64()
    lib.isaac_gpu_sorted_reference_format(ca, C.c_uint32(len(contigs)), ma, C.c_uint32(len(masks)), None, C.c_uint64(0), C.byref(n))
    out = C.create_string_buffer(n.value + 1)
    rc = lib.isaac_gpu_sorted_reference_format(ca, C.c_uint32(len(contigs)), ma, C.c_uint32(len(masks)), out, C.c_uint64(len(out)), C.byref(n))
    if rc:
        raise FormatError(last_error(lib))
    return out.value.decode()
