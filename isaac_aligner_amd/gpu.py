"""Host-side mirror of the reference's interface for the hot path, over the C ABI of include/isaac_gpu.h.

`Aligner.find_matches` stands where alignWorkflow::FindMatchesTransition::perform drives MatchFinder
(lib/workflow/alignWorkflow/FindMatchesTransition.cpp:538-604), `Aligner.select` where SelectMatchesTransition drives
MatchSelector::parallelSelect (lib/alignment/MatchSelector.cpp:370-443).  torch is used for device memory and the stream.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import abi, build as _build

_lib = None


class IsaacGpuError(RuntimeError):
    code = 0          # the ISAAC_GPU_E* value
    error_offset = 0  # isaac_gpu_fastq_to_bcl: byte offset of the malformed record
    n_clusters = 0    # isaac_gpu_fastq_to_bcl: records converted before it


def load_library(path=None):
    """dlopens libisaac_gpu.so; never falls back to anything else"""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- first: the library must bind to the HIP runtime torch ships, not to a second copy
        path = path or os.environ.get("ISAAC_GPU_LIBRARY") or _build.LIB   # the variable selects another build of the same library
        if not os.path.exists(path):
            raise IsaacGpuError("libisaac_gpu.so is missing: run `python -m isaac_aligner_amd.build` (hipcc --offload-arch=gfx950)")
        lib = C.CDLL(path)
        lib.isaac_gpu_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


EXPORTS = ["isaac_gpu_last_error", "isaac_gpu_create", "isaac_gpu_destroy", "isaac_gpu_set_params", "isaac_gpu_index_dev", "isaac_gpu_set_index_dev", "isaac_gpu_malloc", "isaac_gpu_free", "isaac_gpu_upload", "isaac_gpu_download", "isaac_gpu_memory_info", "isaac_gpu_host_malloc", "isaac_gpu_host_free",
           "isaac_gpu_copy", "isaac_gpu_synchronize", "isaac_gpu_set_deferred_completion", "isaac_gpu_load_contigs", "isaac_gpu_load_contigs_dev", "isaac_gpu_load_index", "isaac_gpu_build_index", "isaac_gpu_get_index", "isaac_gpu_get_index_range", "isaac_gpu_get_mask_offsets",
           "isaac_gpu_sorted_reference_parse", "isaac_gpu_sorted_reference_format", "isaac_gpu_sorted_reference_last_error", "isaac_gpu_load_sorted_reference",
           "isaac_gpu_save_sorted_reference",
           "isaac_gpu_find_matches", "isaac_gpu_set_loaded_contigs", "isaac_gpu_build_fragments", "isaac_gpu_determine_tls", "isaac_gpu_share_index", "isaac_gpu_select", "isaac_gpu_select_n", "isaac_gpu_resolve_flagged", "isaac_gpu_set_host_contigs", "isaac_gpu_download_async", "isaac_gpu_download_wait", "isaac_gpu_share_reference", "isaac_gpu_select_candidates",
           "isaac_gpu_bsw_batch", "isaac_gpu_compact_cigars", "isaac_gpu_compact_cigars_async",
           "isaac_gpu_bam_records", "isaac_gpu_bin_tile", "isaac_gpu_bin_tile_map", "isaac_gpu_bam_last_error", "isaac_gpu_bam_header", "isaac_gpu_bgzf_bound", "isaac_gpu_bgzf_compress", "isaac_gpu_bgzf_store_bound", "isaac_gpu_bgzf_store", "isaac_gpu_bgzf_deflate_bound", "isaac_gpu_bgzf_deflate", "isaac_gpu_bam_index", "isaac_gpu_bam_indexer_create", "isaac_gpu_bam_indexer_add", "isaac_gpu_bam_indexer_add_entries", "isaac_gpu_bam_indexer_finish", "isaac_gpu_bam_indexer_destroy", "isaac_gpu_bam_index_last_error",
           "isaac_gpu_default_params", "isaac_gpu_parse_gap_scoring", "isaac_gpu_parse_seeds", "isaac_gpu_parse_adapters", "isaac_gpu_params_last_error",
           "isaac_gpu_fastq_to_bcl", "isaac_gpu_fastq_tile_clusters_max", "isaac_gpu_fastq_tiles", "isaac_gpu_get_counters", "isaac_gpu_kernel_time_ms", "isaac_gpu_reset_timers"]


def fastq_tiles(clusters_loaded, n_seeds, clusters_at_a_time=0, first_tile=1):
    """FastqSeedSource::discoverTiles' breakdown of one load: ([(tile number, clusters)], next tile number)"""
    lib = load_library()
    n, nxt = C.c_uint32(), C.c_uint32()
    lib.isaac_gpu_fastq_tiles(C.c_uint32(clusters_loaded), C.c_uint32(clusters_at_a_time), C.c_uint32(n_seeds), C.c_uint32(first_tile), None, None, C.c_uint32(0), C.byref(n), C.byref(nxt))
    numbers, sizes = (C.c_uint32 * max(1, n.value))(), (C.c_uint32 * max(1, n.value))()
    rc = lib.isaac_gpu_fastq_tiles(C.c_uint32(clusters_loaded), C.c_uint32(clusters_at_a_time), C.c_uint32(n_seeds), C.c_uint32(first_tile), numbers, sizes, C.c_uint32(n.value), C.byref(n), C.byref(nxt))
    if rc:
        raise IsaacGpuError("isaac_gpu_fastq_tiles: %d" % rc)
    return [(numbers[i], sizes[i]) for i in range(n.value)], nxt.value


def _p(t):
    """device (or host) pointer of a torch tensor / numpy array / None"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data_as(C.c_void_p)
    return C.c_void_p(t.data_ptr())


class Aligner:
    """one context on one device"""

    def __init__(self, params, device=0, contigs=None, use_current_stream=True, deferred_completion=False):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise IsaacGpuError("no HIP device is visible: this package has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.params = params
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_current_stream else 0
        h = C.c_void_p()
        self._check(self.lib.isaac_gpu_create(int(device), C.byref(params), C.c_void_p(stream), C.byref(h)))
        self.h = h
        self.n_reads = params.n_reads
        self.cluster_length = params.read_length[0] + params.read_length[1]
        self.n_contigs = 0
        self._inflight = []          # tensors of select calls whose last pass may still be running (deferred completion)
        self._borrowers = weakref.WeakSet()   # contexts that read this one's table through index_tensors() / set_index_tensors()
        self._lender = None
        self.deferred_completion = bool(deferred_completion)
        if self.deferred_completion:
            self._check(self.lib.isaac_gpu_set_deferred_completion(self.h, 1))
        if contigs is not None:
            self.load_contigs(contigs)

    def _refuse_while_lent(self, what):
        open_borrowers = [b for b in getattr(self, "_borrowers", ()) if getattr(b, "h", None)]
        if open_borrowers:
            raise IsaacGpuError("%s: %d other context(s) still read this context's table (index_tensors / set_index_tensors); close them first" % (what, len(open_borrowers)))

    def close(self):
        if getattr(self, "h", None):
            self._refuse_while_lent("close")
            self.lib.isaac_gpu_destroy(self.h)
            self.h = None
            if getattr(self, "_lender", None) is not None:
                self._lender._borrowers.discard(self)
                self._lender = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise IsaacGpuError("isaac_gpu error %d: %s" % (rc, self.lib.isaac_gpu_last_error().decode()))

    # ---- reference ------------------------------------------------------------------------------------------------
    def load_contigs(self, contigs):
        """contigs: list of bytes / uint8 tensors (ASCII ACGTN).  Device tensors are used in place."""
        torch = self.torch
        if getattr(contigs, "padded", None) is not None and contigs.padded.device == self.device:     # synth.Genome already resident: used in place
            offsets = np.asarray(contigs.offsets, np.uint64)
            self._bases, self.contig_offsets, self.n_contigs = contigs.padded, offsets, len(offsets) - 1
            self._check(self.lib.isaac_gpu_load_contigs_dev(self.h, _p(self._bases), _p(offsets), C.c_uint32(self.n_contigs)))
            return
        lengths = [len(c) if isinstance(c, (bytes, bytearray)) else c.numel() for c in contigs]
        offsets = np.zeros(len(contigs) + 1, np.uint64)
        offsets[1:] = np.cumsum(lengths)
        parts = [torch.frombuffer(bytearray(c), dtype=torch.uint8) if isinstance(c, (bytes, bytearray)) else c for c in contigs]
        bases = torch.cat([p.to(self.device) for p in parts] + [torch.full((64,), 78, dtype=torch.uint8, device=self.device)])
        self._bases = bases
        self.contig_offsets = offsets
        self.n_contigs = len(contigs)
        self._check(self.lib.isaac_gpu_load_contigs_dev(self.h, _p(bases), _p(offsets), C.c_uint32(self.n_contigs)))

    def build_index(self, repeat_threshold=1000, annotate_neighbors=True):
        self._refuse_while_lent("build_index")
        n = C.c_uint64()
        self._check(self.lib.isaac_gpu_build_index(self.h, C.c_uint32(repeat_threshold), int(annotate_neighbors), C.byref(n)))
        return n.value

    def load_index(self, masks, karyotype=None):
        """masks: list of REFERENCE_KMER_DTYPE arrays (the mask files in mask order; np.memmap works);
        karyotype: SortedReferenceMetadata::Contig::karyotypeIndex_ per stored contig index, None = identity"""
        self._refuse_while_lent("load_index")
        masks = [np.ascontiguousarray(m, abi.REFERENCE_KMER_DTYPE) for m in masks]
        ptrs = (C.c_void_p * len(masks))(*[m.ctypes.data for m in masks])
        sizes = (C.c_uint64 * len(masks))(*[len(m) for m in masks])
        kar = np.ascontiguousarray(karyotype, np.uint32) if karyotype is not None else None
        self._check(self.lib.isaac_gpu_load_index(self.h, ptrs, sizes, C.c_uint32(len(masks)), _p(kar), C.c_uint32(self.n_contigs)))

    def load_sorted_reference(self, xml_path):
        """isaac-align -r: the mask files named by sorted-reference.xml, with its karyotype translation"""
        from . import sorted_reference
        self._refuse_while_lent("load_sorted_reference")
        rc = self.lib.isaac_gpu_load_sorted_reference(self.h, os.fsencode(xml_path))
        if rc:
            raise IsaacGpuError("isaac_gpu error %d: %s" % (rc, sorted_reference.last_error(self.lib)))

    def save_sorted_reference(self, directory, genome_name, contigs):
        """isaac-sort-reference's output for the resident table; contigs: list of sorted_reference.Contig"""
        from . import sorted_reference
        arr = (sorted_reference.Contig * max(1, len(contigs)))(*contigs)
        rc = self.lib.isaac_gpu_save_sorted_reference(self.h, os.fsencode(directory), genome_name.encode(), arr, C.c_uint32(len(contigs)))
        if rc:
            raise IsaacGpuError("isaac_gpu error %d: %s" % (rc, sorted_reference.last_error(self.lib)))

    def mask_offsets(self, n_masks=64):
        out = np.zeros(n_masks + 1, np.uint64)
        self._check(self.lib.isaac_gpu_get_mask_offsets(self.h, _p(out), C.c_uint32(n_masks)))
        return out

    def get_index(self):
        n = C.c_uint64()
        self._check(self.lib.isaac_gpu_get_index(self.h, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros(n.value, abi.REFERENCE_KMER_DTYPE)
        self._check(self.lib.isaac_gpu_get_index(self.h, _p(out), C.c_uint64(n.value), C.byref(n)))
        return out

    def set_params(self, params):
        """other options / read lengths against the same resident reference and table"""
        self._check(self.lib.isaac_gpu_set_params(self.h, C.byref(params)))
        self.params = params
        self.n_reads = params.n_reads
        self.cluster_length = params.read_length[0] + params.read_length[1]

    def index_tensors(self):
        """the resident table as an [n, 2] int64 device tensor (k-mer, position per entry: the records of the mask files) that aliases the library's memory: no copy"""
        k, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.isaac_gpu_index_dev(self.h, C.byref(k), C.byref(n)))

        owner = self

        class _View:                      # __cuda_array_interface__: torch wraps the pointer without copying
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count, 2), "typestr": "<i8", "data": (ptr, False), "version": 2}
                self.owner = owner        # the tensor aliases the owner's table: it keeps it alive (torch keeps the view object referenced)
        if not n.value:
            return self.torch.empty((0, 2), dtype=self.torch.int64, device=self.device)
        entries = self.torch.as_tensor(_View(k.value, n.value), device=self.device)
        entries._isaac_owner = self      # set_index_tensors registers the borrower with the owner
        return entries

    def set_index_tensors(self, entries, mask_offsets=None):
        """adopts a table held in an [n, 2] int64 device tensor (kept referenced); mask_offsets: the cuts of the mask files or None"""
        assert entries.dtype == self.torch.int64 and entries.dim() == 2 and entries.shape[1] == 2 and entries.is_contiguous()
        self._borrowed_index = entries
        owner = getattr(entries, "_isaac_owner", None)
        if owner is not None and owner is not self:
            owner._borrowers.add(self)                # the owner refuses to rebuild, reload or close its table while a borrower is open
            self._lender = owner
        mo = np.ascontiguousarray(mask_offsets, np.uint64) if mask_offsets is not None else None
        self._check(self.lib.isaac_gpu_set_index_dev(self.h, _p(entries), C.c_uint64(entries.shape[0]), _p(mo), C.c_uint32(len(mo) - 1 if mo is not None else 0)))

    def share_reference(self, owner):
        """isaac_gpu_share_reference: contigs and table of `owner` (another Aligner), in place on one device, copied over the link between two"""
        assert owner is not self and owner.h
        self._check(self.lib.isaac_gpu_share_reference(self.h, owner.h))
        self._bases, self.contig_offsets, self.n_contigs = getattr(owner, "_bases", None), owner.contig_offsets, owner.n_contigs
        owner._borrowers.add(self)                    # the owner refuses to rebuild, reload or close while this context is open
        self._lender = owner

    def resolve_flagged(self, bcl, matches, offsets, tls, records, cigars, tile=0):
        """isaac_gpu_resolve_flagged: the clusters flagged 'MAPQ near an integer' redone on the host with glibc and patched in place; returns (flagged, changed)"""
        n_flagged, n_changed = C.c_uint64(), C.c_uint64()
        self._check(self.lib.isaac_gpu_resolve_flagged(self.h, _p(bcl), C.c_uint32(bcl.shape[0]), C.c_uint32(tile), _p(matches), _p(offsets), C.byref(tls), _p(records), _p(cigars),
                                                       C.byref(n_flagged), C.byref(n_changed)))
        return n_flagged.value, n_changed.value

    def set_loaded_contigs(self, loaded):
        loaded = np.ascontiguousarray(loaded, np.uint8) if loaded is not None else None
        self._check(self.lib.isaac_gpu_set_loaded_contigs(self.h, _p(loaded), C.c_uint32(self.n_contigs)))

    # ---- find -----------------------------------------------------------------------------------------------------
    def match_capacity(self, n_clusters):
        """match records isaac_gpu_find_matches is given room for by default (it reports the number it needs when that is not enough)"""
        per = 2 * self.params.n_seeds * max(1, self.params.repeat_threshold - 1)
        return max(1, min(n_clusters * per, max(1024, n_clusters * 24)))

    def find_matches(self, bcl, tile=0, capacity=None, out=None):
        """bcl: uint8 device tensor [n_clusters, cluster_length].  Returns (matches tensor [n,2] int64, offsets tensor, contig_has_matches).
        out: (matches [capacity, 2] int64, offsets [n_clusters + 1] int64) tensors to write into instead of fresh ones"""
        torch = self.torch
        n_clusters = bcl.shape[0]
        per = 2 * self.params.n_seeds * max(1, self.params.repeat_threshold - 1)
        capacity = max(1, capacity or self.match_capacity(n_clusters))
        while True:
            if out is not None and out[0].shape[0] >= capacity and out[1].shape[0] >= n_clusters + 1:
                matches, offsets = out[0], out[1][:n_clusters + 1]
                capacity = matches.shape[0]
            else:
                matches = torch.empty((capacity, 2), dtype=torch.int64, device=self.device)
                offsets = torch.empty(n_clusters + 1, dtype=torch.int64, device=self.device)
            hits = np.zeros(self.n_contigs, np.uint8)
            n = C.c_uint64()
            rc = self.lib.isaac_gpu_find_matches(self.h, _p(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), _p(matches), C.c_uint64(capacity), _p(offsets), C.byref(n), _p(hits))
            if rc == 4 and capacity < n_clusters * per:   # ISAAC_GPU_ECAPACITY
                capacity = min(n_clusters * per, max(n.value, capacity * 2))
                out = None
                continue
            self._check(rc)
            return matches[:n.value], offsets, hits

    # ---- extend ---------------------------------------------------------------------------------------------------
    def build_fragments(self, bcl, matches, offsets, tile=0, with_gaps=True, trim=True, want_output=True):
        torch = self.torch
        n_clusters = bcl.shape[0]
        nc, ng = C.c_uint64(), C.c_uint64()
        if not want_output:
            self._check(self.lib.isaac_gpu_build_fragments(self.h, _p(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), _p(matches), _p(offsets), int(with_gaps), int(trim),
                                                           None, C.c_uint64(0), C.byref(nc), None, C.c_uint64(0), C.byref(ng)))
            return None, None
        cap = n_clusters * 8 + 1024
        while True:
            cands = torch.empty(cap * abi.CANDIDATE_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
            cig = torch.empty(cap * 6, dtype=torch.int32, device=self.device)
            rc = self.lib.isaac_gpu_build_fragments(self.h, _p(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), _p(matches), _p(offsets), int(with_gaps), int(trim),
                                                    _p(cands), C.c_uint64(cap), C.byref(nc), _p(cig), C.c_uint64(cap * 6), C.byref(ng))
            if rc == 4:
                cap = max(cap * 2, nc.value + 16, (ng.value + 5) // 6 + 16)
                continue
            self._check(rc)
            c = cands.cpu().numpy().view(abi.CANDIDATE_DTYPE)[:nc.value].copy()
            g = cig.cpu().numpy().view(np.uint32)[:ng.value].copy()
            return c, g

    def determine_tls(self, bcl, matches, offsets, tile=0):
        t = abi.Tls()
        self._check(self.lib.isaac_gpu_determine_tls(self.h, _p(bcl), C.c_uint32(bcl.shape[0]), C.c_uint32(tile), _p(matches), _p(offsets), C.byref(t)))
        return t

    def select(self, bcl, matches, offsets, tls, tile=0, out=None):
        """returns (records tensor uint8 [n_clusters*n_reads, 64], cigar tensor int32)"""
        torch = self.torch
        n_clusters = bcl.shape[0]
        n_rec = n_clusters * self.n_reads
        if out is None:
            records = torch.empty((n_rec, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
            cigars = torch.empty(n_rec * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=self.device)
        else:
            records, cigars = out
        # matches is the slice find_matches returned: its length is the tile's match count, which spares the call a host wait
        self._check(self.lib.isaac_gpu_select_n(self.h, _p(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), _p(matches), C.c_uint64(matches.shape[0]), _p(offsets), C.byref(tls),
                                                _p(records), _p(cigars), C.c_uint64(cigars.numel())))
        if self.deferred_completion:
            # the call's kernels may still be queued: the tensors stay referenced until synchronize()
            self._inflight.append((bcl, matches, offsets, records, cigars))
        return records, cigars

    def align_tile(self, bcl, tile=0, tls=None):
        """find_matches -> determine_tls (unless given) -> select for one tile; returns (records, cigars)"""
        matches, offsets, _ = self.find_matches(bcl, tile=tile)
        if tls is None:
            tls = self.determine_tls(bcl, matches, offsets, tile=tile)
        return self.select(bcl, matches, offsets, tls, tile=tile)

    def select_candidates(self, bcl, candidates, candidate_cigars, tls, tile=0):
        """TemplateBuilder::buildTemplate on explicit candidate lists.  candidates: abi.CANDIDATE_DTYPE numpy array ordered by
        (cluster, read, list position); returns (records tensor, cigars tensor) like select()"""
        torch = self.torch
        n = int(bcl.shape[0])
        cands = np.ascontiguousarray(candidates, abi.CANDIDATE_DTYPE)
        counts = np.bincount(cands["cluster"].astype(np.int64), minlength=n) if len(cands) else np.zeros(n, np.int64)
        offsets = np.zeros(n + 1, np.uint64); offsets[1:] = np.cumsum(counts)
        cand_d = torch.from_numpy(cands.view(np.uint8).reshape(-1).copy() if len(cands) else np.zeros(abi.CANDIDATE_DTYPE.itemsize, np.uint8)).to(self.device)
        off_d = torch.from_numpy(offsets.view(np.int64)).to(self.device)
        cig_in = torch.from_numpy(np.ascontiguousarray(candidate_cigars, np.uint32).view(np.int32) if len(candidate_cigars) else np.zeros(1, np.int32)).to(self.device)
        n_rec = n * self.n_reads
        records = torch.empty((n_rec, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        cigars = torch.empty(n_rec * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=self.device)
        self._check(self.lib.isaac_gpu_select_candidates(self.h, _p(bcl), C.c_uint32(n), C.c_uint32(tile), _p(cand_d), _p(off_d), _p(cig_in), C.byref(tls),
                                                         _p(records), _p(cigars), C.c_uint64(cigars.numel())))
        return records, cigars

    def compact_cigars(self, records, cigars, out=None):
        """packs the 40-word CIGAR slots of select() back to back; rewrites the records' cigar_offset in place.
        Returns (packed cigar tensor view, number of words)"""
        n_rec = records.shape[0]
        if out is None:
            out = self.torch.empty(max(1, n_rec * 4), dtype=self.torch.int32, device=self.device)
        n = C.c_uint64()
        rc = self.lib.isaac_gpu_compact_cigars(self.h, _p(records), C.c_uint64(n_rec), _p(cigars), _p(out), C.c_uint64(out.numel()), C.byref(n))
        if rc == 4:
            out = self.torch.empty(n.value, dtype=self.torch.int32, device=self.device)
            rc = self.lib.isaac_gpu_compact_cigars(self.h, _p(records), C.c_uint64(n_rec), _p(cigars), _p(out), C.c_uint64(out.numel()), C.byref(n))
        self._check(rc)
        return out[:n.value], n.value

    def compact_cigars_async(self, records, cigars, out, n_words_dev):
        """compact_cigars without the host wait: n_words_dev (1-element int64 device tensor) receives the packed length behind the kernels;
        `out` must be large enough (a pool that is too small leaves everything as it was and shows as n_words_dev > out.numel())"""
        self._check(self.lib.isaac_gpu_compact_cigars_async(self.h, _p(records), C.c_uint64(records.shape[0]), _p(cigars), _p(out), C.c_uint64(out.numel()), _p(n_words_dev)))
        if self.deferred_completion:
            self._inflight.append((records, cigars, out, n_words_dev))

    # ---- output format --------------------------------------------------------------------------------------------
    def bin_tile(self, bcl, records, cigars, bin_of_contig, n_bins, cut_positions=None):
        """isaac_gpu_bin_tile / isaac_gpu_bin_tile_map (with cut_positions: ascending ReferencePosition values at which a contig goes on into its next bin):
        the output of one select() cut into a compact tile per bin; returns [(bcl, records, cigars)] per bin (device tensors, views of one buffer)"""
        from . import bam
        torch = self.torch
        n_clusters = bcl.shape[0]
        boc = np.ascontiguousarray(bin_of_contig, np.uint32)
        sizes = (bam.BinSize * n_bins)()
        need = C.c_uint64()
        out = torch.empty(max(64, int(1.2 * (bcl.numel() + records.numel() + 8 * records.shape[0])) + 256 * n_bins), dtype=torch.uint8, device=self.device)
        if cut_positions is None:
            args = lambda buf: (self.h, _p(bcl), _p(records), _p(cigars), C.c_uint32(n_clusters), _p(boc), C.c_uint32(len(boc)), C.c_uint32(n_bins), _p(buf), C.c_uint64(buf.numel()), sizes, C.byref(need))
            entry = self.lib.isaac_gpu_bin_tile
        else:
            cuts = np.ascontiguousarray(cut_positions, np.uint64)
            bin_map = bam.BinMap(_p(boc), len(boc), _p(cuts), len(cuts), n_bins)
            args = lambda buf: (self.h, _p(bcl), _p(records), _p(cigars), C.c_uint32(n_clusters), C.byref(bin_map), _p(buf), C.c_uint64(buf.numel()), sizes, C.byref(need))
            entry = self.lib.isaac_gpu_bin_tile_map
        rc = entry(*args(out))
        if rc == 4:
            out = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            rc = entry(*args(out))
        self._check(rc)
        parts, at = [], 0
        align = lambda v: (v + 63) & ~63
        for b in range(n_bins):
            m, w = sizes[b].n_clusters, sizes[b].n_cigar_words
            bcl_at = at
            rec_at = align(bcl_at + m * self.cluster_length)
            cig_at = align(rec_at + m * self.n_reads * abi.FRAGMENT_DTYPE.itemsize)
            at = align(cig_at + 4 * w)
            parts.append((out[bcl_at:bcl_at + m * self.cluster_length].view(m, self.cluster_length) if m else out[:0].view(0, self.cluster_length),
                          out[rec_at:rec_at + m * self.n_reads * abi.FRAGMENT_DTYPE.itemsize].view(m * self.n_reads, abi.FRAGMENT_DTYPE.itemsize),
                          out[cig_at:cig_at + 4 * w].view(torch.int32)))
        return parts

    def bam_records(self, tiles, out=None, read_group=None, barcode=None, forced_dodgy_alignment_score=None, pessimistic_mapq=False, mark_duplicates=False, keep_duplicates=True, realign_gaps=False, tls=None,
                    bin_contigs=None, bin_unaligned=False, bin_positions=None, realign_vigorously=False):
        """build::Build's BAM alignment records of one or more tiles, in file order (mark_duplicates / keep_duplicates / realign_gaps: BinSorter's steps before the order).
        tiles: [(bcl, records, cigars, read_name_prefix[, read_group[, tls]])] as given to / returned by select().  Returns (uint8 device tensor of the
        uncompressed records, number of records, offset of the unaligned bin)."""
        from . import bam
        arr = (bam.BamTile * len(tiles))()
        keep = []
        n_rec = 0
        for i, tile in enumerate(tiles):
            bcl, records, cigars, prefix = tile[:4]
            name = prefix.encode() if isinstance(prefix, str) else prefix
            keep.append(name)
            arr[i].bcl_dev = _p(bcl).value if _p(bcl) is not None else None
            arr[i].fragments_dev = _p(records).value if _p(records) is not None else None
            arr[i].cigar_dev = _p(cigars).value if _p(cigars) is not None else None
            arr[i].n_records = records.shape[0]
            arr[i].read_name_prefix = name
            if len(tile) > 4 and tile[4] is not None:
                keep.append(tile[4].encode())
                arr[i].read_group = keep[-1]
            if len(tile) > 5 and tile[5] is not None:
                arr[i].tls = C.cast(C.pointer(tile[5]), C.c_void_p)
            n_rec += records.shape[0]
        options = None
        if read_group is not None or barcode is not None or forced_dodgy_alignment_score is not None or pessimistic_mapq or mark_duplicates or not keep_duplicates or realign_gaps or bin_contigs is not None or bin_unaligned or bin_positions is not None:
            options = bam.BamOptions()
            if bin_positions is not None:                       # one bin of the file: the positions [first, end) as ReferencePosition values
                options.bin_filter = 2
                options.bin_first_position, options.bin_end_position = bin_positions
                options.bin_unaligned = int(bool(bin_unaligned))
            elif bin_contigs is not None or bin_unaligned:      # one bin of the file: contigs [first, end) and / or the templates without a position
                options.bin_filter = 1
                options.bin_first_contig, options.bin_end_contig = bin_contigs if bin_contigs is not None else (0, 0)
                options.bin_unaligned = int(bool(bin_unaligned))
            options.mark_duplicates, options.keep_duplicates = int(bool(mark_duplicates)), int(bool(keep_duplicates))
            options.realign_gaps = int(bool(realign_gaps))
            options.realign_vigorously = int(bool(realign_vigorously))
            options.tls = C.cast(C.pointer(tls), C.c_void_p) if tls is not None else None
            options.forced_dodgy_alignment_score = (self.params.dodgy_alignment_score & 0xff) if forced_dodgy_alignment_score is None else forced_dodgy_alignment_score
            options.pessimistic_mapq = int(bool(pessimistic_mapq))
            options.read_group = None if read_group is None else read_group.encode()
            options.barcode = None if barcode is None else barcode.encode()
        if out is None:
            out = self.torch.empty(max(1, n_rec * (96 + 2 * max(self.params.read_length[0], self.params.read_length[1]))), dtype=self.torch.uint8, device=self.device)
        nb, nr, un = C.c_uint64(), C.c_uint64(), C.c_uint64()

        def call(buf):
            return self.lib.isaac_gpu_bam_records(self.h, arr, C.c_uint32(len(tiles)), C.byref(options) if options is not None else None, _p(buf), C.c_uint64(buf.numel()),
                                                  C.byref(nb), C.byref(nr), C.byref(un))
        rc = call(out)
        if rc == 4:
            out = self.torch.empty(nb.value, dtype=self.torch.uint8, device=self.device)
            rc = call(out)
        self._check(rc)
        return out[:nb.value], nr.value, un.value

    def bgzf_store(self, data, eof_block=False, out=None):
        """BGZF framing without compression (--bam-gzip-level 0) of a uint8 device tensor, CRC-32 on the device; returns a uint8 device tensor"""
        self.lib.isaac_gpu_bgzf_store_bound.restype = C.c_uint64
        bound = self.lib.isaac_gpu_bgzf_store_bound(C.c_uint64(data.numel()))
        if out is None:
            out = self.torch.empty(bound, dtype=self.torch.uint8, device=self.device)
        n = C.c_uint64()
        self._check(self.lib.isaac_gpu_bgzf_store(self.h, _p(data), C.c_uint64(data.numel()), C.c_int(int(eof_block)), _p(out), C.c_uint64(out.numel()), C.byref(n)))
        return out[:n.value]

    def bgzf_deflate(self, data, eof_block=False, out=None):
        """BGZF with compression on the device (--bam-gzip-level 1) of a uint8 device tensor; returns a uint8 device tensor (a view of `out`)"""
        self.lib.isaac_gpu_bgzf_deflate_bound.restype = C.c_uint64
        if out is None:
            out = self.torch.empty(self.lib.isaac_gpu_bgzf_deflate_bound(C.c_uint64(data.numel())), dtype=self.torch.uint8, device=self.device)
        n = C.c_uint64()
        self._check(self.lib.isaac_gpu_bgzf_deflate(self.h, _p(data), C.c_uint64(data.numel()), C.c_int(int(eof_block)), _p(out), C.c_uint64(out.numel()), C.byref(n)))
        return out[:n.value]

    def records_to_numpy(self, records, cigars):
        if self.deferred_completion:
            self.synchronize()
        return records.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1).copy(), cigars.cpu().numpy().view(np.uint32).copy()

    # ---- input format ---------------------------------------------------------------------------------------------
    def fastq_to_bcl(self, text, read_index, bcl=None, max_clusters=None, allow_variable_length=False, final=True):
        """io::FastqReader / FastqLoader::loadSingleRead on the device.  text: bytes or a uint8 device tensor of FASTQ text.
        Returns (bcl tensor [max_clusters, cluster_length], n_clusters, consumed_bytes); raises IsaacGpuError for malformed input."""
        torch = self.torch
        if isinstance(text, (bytes, bytearray)):
            text = torch.frombuffer(bytearray(text) + bytearray(16), dtype=torch.uint8)[:len(text)].to(self.device)
        n = int(text.numel())
        if max_clusters is None:
            max_clusters = n // 4 + 1 if bcl is None else int(bcl.shape[0])
        if bcl is None:
            bcl = torch.zeros((max_clusters, self.cluster_length), dtype=torch.uint8, device=self.device)
        nc, consumed, err = C.c_uint32(), C.c_uint64(), C.c_uint64()
        rc = self.lib.isaac_gpu_fastq_to_bcl(self.h, _p(text), C.c_uint64(n), C.c_uint32(read_index), int(allow_variable_length), int(final), _p(bcl),
                                             C.c_uint32(max_clusters), C.byref(nc), C.byref(consumed), C.byref(err))
        if rc:
            e = IsaacGpuError("isaac_gpu error %d: %s" % (rc, self.lib.isaac_gpu_last_error().decode()))
            e.code, e.error_offset, e.n_clusters = rc, err.value, nc.value
            raise e
        return bcl, nc.value, consumed.value

    # ---- leaf -----------------------------------------------------------------------------------------------------
    def bsw_batch(self, scores, queries, databases):
        """scores = (match, mismatch, gapOpen > 0, gapExtend > 0) as BandedSmithWaterman's constructor takes them"""
        torch = self.torch
        blob = bytearray()
        jobs = np.zeros(len(queries), abi.BSW_JOB_DTYPE)
        for i, (q, d) in enumerate(zip(queries, databases)):
            q = q.encode() if isinstance(q, str) else bytes(q)
            d = d.encode() if isinstance(d, str) else bytes(d)
            assert len(d) == len(q) + 15
            jobs[i] = (len(blob), len(blob) + len(q), len(q), 0)
            blob += q + d
        seq = torch.frombuffer(blob + bytearray(32), dtype=torch.uint8).to(self.device)
        jobs_d = torch.from_numpy(jobs.view(np.uint8)).to(self.device)
        res = torch.zeros(len(queries) * abi.BSW_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        max_len = int(jobs["query_length"].max()) if len(jobs) else 1
        self._check(self.lib.isaac_gpu_bsw_batch(self.h, int(scores[0]), int(scores[1]), int(scores[2]), int(scores[3]), _p(seq), _p(jobs_d), C.c_uint32(len(queries)),
                                                 C.c_uint32(max_len), _p(res)))
        self.synchronize()
        return res.cpu().numpy().view(abi.BSW_RESULT_DTYPE)

    # ---- bookkeeping ----------------------------------------------------------------------------------------------
    def synchronize(self):
        self._check(self.lib.isaac_gpu_synchronize(self.h))
        self._inflight.clear()

    def counters(self):
        c = abi.Counters()
        self._check(self.lib.isaac_gpu_get_counters(self.h, C.byref(c)))
        return c.asdict()

    def kernel_time_ms(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.isaac_gpu_kernel_time_ms(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_timers(self):
        self._check(self.lib.isaac_gpu_reset_timers(self.h))
