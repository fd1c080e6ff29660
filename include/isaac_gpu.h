/*
 * isaac_gpu.h -- C ABI of the MI355X (gfx950) implementation of Isaac's seed-and-extend hot path.
 *
 * The reference (sequencing/isaac_aligner, iSAAC-01.15.04.01) has no plugin/FFI seam: the path is reached by direct C++
 * calls inside one static binary.  The entry points below replace the narrowest C++ seams of that path; each one cites
 * the reference interface it stands in for (paths relative to /root/reference/src/c++).  A maintainer binds them from
 * the two workflow transitions.
 *
 * Conventions
 *   - plain C, opaque context, caller-owned buffers, no exceptions across the boundary: every call returns 0 on success
 *     or an ISAAC_GPU_E* code; isaac_gpu_last_error() gives the text.
 *   - one context per device; calls on one context are serialised by the caller, different contexts are independent.
 *   - bulk buffers (`*_dev` parameters) are DEVICE pointers (hipMalloc'ed or owned by a framework that shares the HIP
 *     context, e.g. torch tensors); everything else is host memory.  isaac_gpu_malloc/upload/download exist so that a
 *     host with no other GPU runtime can use the library.
 *   - all kernels run on the stream given at context creation (0 = the default stream).
 *   - there is no CPU fallback: without a usable HIP device isaac_gpu_create fails.
 */
#ifndef ISAAC_GPU_H
#define ISAAC_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISAAC_GPU_OK 0
#define ISAAC_GPU_EINVAL 1     /* bad argument (where the reference throws PreConditionException / InvalidOptionException) */
#define ISAAC_GPU_ENOMEM 2     /* device allocation failed (reference: common::MemoryException) */
#define ISAAC_GPU_EHIP 3       /* HIP runtime error */
#define ISAAC_GPU_ECAPACITY 4  /* an output buffer supplied by the caller is too small */
#define ISAAC_GPU_EOVERFLOW 5  /* an internal fixed-capacity work list overflowed; results of the flagged clusters are not exact */
#define ISAAC_GPU_EFORMAT 6    /* malformed FASTQ (reference: io::FastqFormatException) */
#define ISAAC_GPU_EREADLEN 7   /* FASTQ read shorter than the configured read length (reference: common::IoException, EINVAL) */

#define ISAAC_GPU_MAX_SEEDS 16
#define ISAAC_GPU_MAX_CIGAR_OPS 40
#define ISAAC_GPU_MAX_ADAPTERS 8

/* alignment::SeedMetadata (include/alignment/SeedMetadata.hh:43-101): index in the list == seed index of SeedId */
typedef struct { uint16_t offset, length; uint32_t read_index; } isaac_seed;

/* flowcell::SequencingAdapterMetadata (include/flowcell/SequencingAdapterMetadata.hh:33-73): one adapter of --default-adapters, written in the direction of the
 * reference (5..126 bases of ACGT, NUL-terminated); reverse: the strand it is expected on ("*ACGT" forms); clip_length: 0 for the unbounded forms ("ACGT*",
 * "*ACGT": everything from the adapter to the read's end goes), the adapter's length for plain sequences. */
typedef struct { char sequence[128]; uint32_t reverse; uint32_t clip_length; } isaac_adapter;

/* The subset of options::AlignOptions (lib/options/AlignOptions.cpp:77-160) that parameterises the path, plus the read
 * geometry of flowcell::ReadMetadataList and the seed list of --seeds (alignOptions/SeedDescriptorOption.cpp:90-151). */
typedef struct
{
    int32_t gap_match, gap_mismatch, gap_open, gap_extend, min_gap_extend; /* --gap-scoring, bwa = 0:-3:-11:-4:-20 */
    uint32_t repeat_threshold;         /* --repeat-threshold 10 */
    uint32_t gapped_mismatches_max;    /* --gapped-mismatches 5 */
    uint32_t semialigned_gap_limit;    /* --semialigned-gap-limit 100 */
    uint32_t base_quality_cutoff;      /* --base-quality-cutoff 25 */
    uint32_t ignore_neighbors;         /* --ignore-neighbors 0 */
    uint32_t clip_semialigned;         /* --clip-semialigned 1 */
    uint32_t clip_overlapping;         /* --clip-overlapping 1 */
    uint32_t scatter_repeats;          /* --scatter-repeats 0 */
    int32_t dodgy_alignment_score;     /* --dodgy-alignment-score 0 (255 = Unknown, -1 = Unaligned) */
    uint32_t mapq_threshold;           /* --mapq-threshold 0 */
    uint32_t keep_unaligned;           /* --keep-unaligned back|front => 1 */
    int32_t mate_drift_range;          /* --shadow-scan-range -1 */
    uint32_t first_pass_seeds;         /* --first-pass-seeds (2 with --seeds auto and a non-zero gap limit) */
    uint32_t seed_length;              /* --seed-length: only 32 is implemented */
    uint32_t n_reads;                  /* 1 or 2 */
    uint32_t read_length[2];
    uint32_t n_seeds;
    isaac_seed seeds[ISAAC_GPU_MAX_SEEDS];
    /* --default-adapters of the flowcell (lib/options/AlignOptions.cpp:189-207; 0: nothing is clipped, the default): what
     * matchSelector::FragmentSequencingAdapterClipper clips by in UngappedAligner, GappedAligner and ShadowAligner (isaac_gpu_parse_adapters fills these) */
    uint32_t n_adapters;
    isaac_adapter adapters[ISAAC_GPU_MAX_ADAPTERS];
} isaac_params;

/* alignment::Match (include/alignment/Match.hh:38-73): SeedId (SeedId.hh:60-127) + ReferencePosition value
 * (ReferencePosition.hh:51-188), exactly the 16 bytes io::TileMatchWriter::write appends (lib/io/MatchWriter.cpp:79-94) */
typedef struct { uint64_t seed_id; uint64_t location; } isaac_match;

/* reference::ReferenceKmer<unsigned long> (include/reference/ReferenceKmer.hh:37-54): the record of the mask files */
typedef struct { uint64_t kmer; uint64_t position; } isaac_reference_kmer;

/* one candidate alignment of one read: the observable fields of alignment::FragmentMetadata
 * (include/alignment/FragmentMetadata.hh:330-414) after FragmentBuilder::build */
typedef struct
{
    int64_t position; double log_probability;
    uint32_t cluster, read_index, contig_id, observed_length, reverse, mismatch_count, matches_in_a_row, gap_count, edit_distance,
             smith_waterman_score, unique_seed_count, non_unique_first, non_unique_second, repeat_seeds_count, cigar_offset, cigar_length,
             low_clipped, high_clipped;
    int32_t first_seed_index; uint32_t reserved;
} isaac_candidate;

/* alignment::TemplateLengthStatistics (include/alignment/TemplateLengthStatistics.hh:44-239) */
typedef struct { uint32_t min, max, median, low_std_dev, high_std_dev; int32_t best_model[2]; uint32_t stable, mate_min, mate_max; } isaac_tls;

/* What the reference persists per read: io::FragmentHeader (include/io/Fragment.hh:73-330), plus the BAM MAPQ that
 * build::FragmentAccessorBamAdapter::mapq() derives from it (include/build/FragmentAccessorBamAdapter.hh:250-265). */
typedef struct
{
    uint64_t f_strand_position, mate_f_strand_position;   /* ReferencePosition values */
    int32_t bam_tlen; uint32_t observed_length;
    uint16_t low_clipped, high_clipped, alignment_score, template_alignment_score;
    uint16_t read_length, cigar_length, gap_count, edit_distance;
    uint32_t flags;        /* bit0 paired,1 unmapped,2 mateUnmapped,3 reverse,4 mateReverse,5 firstRead,6 secondRead,7 failFilter,8 properPair */
    uint32_t cigar_offset; /* into the cigar pool of the same call */
    uint32_t tile, cluster_id;
    uint32_t mapq;
    uint32_t reserved;
} isaac_fragment;

/* one banded Smith-Waterman problem: query = ASCII ACGTn, database = query_length + 15 ASCII ACGTN bytes */
typedef struct { uint64_t query_offset, database_offset; uint32_t query_length; uint32_t reserved; } isaac_bsw_job;
typedef struct { uint32_t n_ops, offset; uint32_t cigar[ISAAC_GPU_MAX_CIGAR_OPS]; } isaac_bsw_result;

/* work counters of the last isaac_gpu_align_tile call; they feed the bytes/pair formula of SURVEY.md §8d */
typedef struct
{
    uint64_t clusters, probes, probe_steps, matches, candidates, ungapped_scans, bsw_jobs, bsw_accepted, simple_indels,
             rescue_calls, rescue_window_bases, rescue_candidates, rescue_bsw, overflow_clusters, mapq_near_integer, heavy_clusters,
             residual_capacity, residual_near_tie, residual_oversize, large_sums;   /* heavy_clusters by cause; clusters whose probability sums took a whole workgroup */
} isaac_counters;

typedef struct isaac_gpu_ctx isaac_gpu_ctx;

const char *isaac_gpu_last_error(void);

/* `stream` of isaac_gpu_create: a hipStream_t of the caller's, NULL for the device's default stream, or this value for a stream of the context's
 * own (created with it, destroyed with it, not synchronised with the default stream): what several contexts on one device want when their
 * callers are threads of a host that has no HIP runtime of its own. */
#define ISAAC_GPU_STREAM_OWN ((void *)(uintptr_t)1)
/* Replaces the construction of alignment::MatchFinder / MatchSelector / TemplateBuilder for one device
 * (lib/alignment/MatchFinder.cpp:74-112, MatchSelector.cpp:92-168).  `stream` is a hipStream_t or NULL. */
int isaac_gpu_create(int device, const isaac_params *params, void *stream, isaac_gpu_ctx **out);
void isaac_gpu_destroy(isaac_gpu_ctx *ctx);
/* Replaces the options and the read geometry of the context; contigs and table stay resident.  Waits for what is in flight. */
int isaac_gpu_set_params(isaac_gpu_ctx *ctx, const isaac_params *params);

/* plain device memory helpers for hosts without another GPU runtime */
int isaac_gpu_malloc(isaac_gpu_ctx *ctx, uint64_t bytes, void **dev_out);
int isaac_gpu_free(isaac_gpu_ctx *ctx, void *dev);
int isaac_gpu_upload(isaac_gpu_ctx *ctx, void *dev, const void *host, uint64_t bytes);
int isaac_gpu_download(isaac_gpu_ctx *ctx, void *host, const void *dev, uint64_t bytes);
/* The same on a copy stream of the context's own, behind everything the context's stream has been given so far, without waiting: *ticket_out names the copy.
 * isaac_gpu_download_wait(ctx, ticket) returns when that copy -- and every one begun before it -- has arrived.  The device buffer must stay untouched and the host
 * buffer unread until then; `host` should be page-locked (isaac_gpu_host_malloc), or the runtime stages the copy and the call waits after all.  What lets a host
 * fetch one bin's blocks while the next bin is being encoded (isaac-align's build stage). */
int isaac_gpu_download_async(isaac_gpu_ctx *ctx, void *host, const void *dev, uint64_t bytes, uint64_t *ticket_out);
int isaac_gpu_download_wait(isaac_gpu_ctx *ctx, uint64_t ticket);
/* page-locked host memory: uploads from it and downloads into it run at the link's rate (pageable memory: a third of it), and nothing clears it first */
int isaac_gpu_host_malloc(uint64_t bytes, void **host_out);
int isaac_gpu_host_free(void *host);
/* free and total memory of the context's device, in bytes (a host that keeps work on the device while there is room: isaac-align's bins) */
int isaac_gpu_memory_info(isaac_gpu_ctx *ctx, uint64_t *free_bytes_out, uint64_t *total_bytes_out);
/* device to device, on the context's stream (ordered with the calls before and after it; no host wait) */
int isaac_gpu_copy(isaac_gpu_ctx *ctx, void *dst_dev, const void *src_dev, uint64_t bytes);
int isaac_gpu_synchronize(isaac_gpu_ctx *ctx);
/* Deferred completion (off by default).  When on, isaac_gpu_select / isaac_gpu_select_candidates return while their last
 * kernels are still running on the context's stream (the call only enqueues; nothing inside it waits for the GPU), so that the
 * host can prepare and enqueue the next call meanwhile.  The caller then owns the hazard: every buffer handed to such a call (bcl, matches, offsets, fragments, cigar) must
 * stay allocated and unmodified, and its outputs unread, until isaac_gpu_synchronize() returns; calls in flight need distinct
 * output buffers.  Every other entry point of the context is ordered behind those kernels on the same stream.  Turning it off completes what is in flight. */
int isaac_gpu_set_deferred_completion(isaac_gpu_ctx *ctx, int enabled);

/* Replaces reference::loadContigs (include/reference/ContigLoader.hh:84-140, lib/reference/ContigLoader.cpp:29-66):
 * ASCII ACGTN, exactly reference::Contig::forward_, all contigs concatenated in karyotype order;
 * offsets has n_contigs + 1 entries.  The bytes are copied to HBM once and stay resident. */
int isaac_gpu_load_contigs(isaac_gpu_ctx *ctx, const char *bases_host, const uint64_t *offsets_host, uint32_t n_contigs);
int isaac_gpu_load_contigs_dev(isaac_gpu_ctx *ctx, const char *bases_dev, const uint64_t *offsets_host, uint32_t n_contigs);

/* Replaces the streaming of the mask files of sorted-reference.xml by matchFinder::ExactMaskMatcher::matchMask
 * (lib/alignment/matchFinder/ExactMaskMatcher.cpp:83-126, MatchFinder.cpp:251-316): the masks (host pointers to the
 * mmap'ed *.dat files, in mask order, which is global k-mer order) are concatenated into one resident sorted table, streamed
 * through device staging buffers 256 MB at a time (no host copy is made; GRCh38: 46 GB).  karyotype_of_contig is
 * SortedReferenceMetadata::Contig::karyotypeIndex_ (MatchFinder.cpp:51-66: stored contig index -> position in karyotype order,
 * which is the order of isaac_gpu_load_contigs), n_contigs entries, NULL = identity.  Tables below 2^32 entries. */
int isaac_gpu_load_index(isaac_gpu_ctx *ctx, const isaac_reference_kmer *const *masks_host, const uint64_t *mask_sizes, uint32_t n_masks,
                         const uint32_t *karyotype_of_contig, uint32_t n_contigs);

/* Replaces isaac-sort-reference (lib/reference/ReferenceSorter.cpp:105-261 + NeighborsFinder.cpp:193-446) for the
 * contigs already loaded: builds the sorted 32-mer table on the device, one mask (top 6 bits of the k-mer, --mask-width 6) after
 * the other exactly as the 64 mask files are laid out (repeat_threshold = 1000, neighborhood 4).  Sized for a human genome:
 * 64-bit positions and counts throughout, below 2^32 table entries.  n_entries_out may be NULL. */
int isaac_gpu_build_index(isaac_gpu_ctx *ctx, uint32_t repeat_threshold, int annotate_neighbors, uint64_t *n_entries_out);
/* copies the resident table back in mask-file record layout (capacity in records) */
int isaac_gpu_get_index(isaac_gpu_ctx *ctx, isaac_reference_kmer *out_host, uint64_t capacity, uint64_t *n_out);
/* entries [first, first + n) of the resident table, e.g. one mask at a time */
int isaac_gpu_get_index_range(isaac_gpu_ctx *ctx, uint64_t first, uint64_t n, isaac_reference_kmer *out_host);
/* offsets_out[m] = table entries before mask m, m = 0 .. n_masks (the <File> elements of sorted-reference.xml,
 * lib/reference/SortedReferenceXml.cpp:312-324: one mask file = entries [offsets[m], offsets[m + 1])); n_masks must be the
 * number of masks the table was loaded or built with (64 for a built one) */
int isaac_gpu_get_mask_offsets(isaac_gpu_ctx *ctx, uint64_t *offsets_out, uint32_t n_masks);

/* The resident table as it lies in HBM: entries_dev[i] = the i-th record of the concatenated mask files, k-mer and position side by side as the files
 * hold them (a read-only device pointer, valid until the table is rebuilt or reloaded or the context destroyed; rounds 1-4 kept two arrays, which cost every
 * probe that finds its k-mer a second cache line).  isaac_gpu_set_index_dev adopts such a
 * table owned by the caller instead of loading one -- several contexts on one device share one table that way, and the ranks of a multi-GPU
 * job can receive rank 0's table over xGMI instead of building or reading their own.  mask_offsets (n_masks + 1 entries) may be NULL
 * when nobody will ask for the mask cuts.  The contig translation of isaac_gpu_load_index does not apply (positions are used as stored). */
int isaac_gpu_index_dev(isaac_gpu_ctx *ctx, const isaac_reference_kmer **entries_dev_out, uint64_t *n_entries_out);
int isaac_gpu_set_index_dev(isaac_gpu_ctx *ctx, const isaac_reference_kmer *entries_dev, uint64_t n_entries, const uint64_t *mask_offsets, uint32_t n_masks);
/* The same for two contexts of one process, whole: `ctx` takes `owner`'s table as it is -- entries, mask cuts, karyotype translation -- in place
 * when both are on one device (nothing is copied; `owner` must outlive `ctx` and keep its table), as a copy over the link between the two devices
 * when they are not.  Both must have loaded the same contigs.  What a host does for the second and further workers of a run (--devices). */
int isaac_gpu_share_index(isaac_gpu_ctx *ctx, isaac_gpu_ctx *owner);
/* Contigs and table together, for a context that has loaded nothing: on one device `ctx` reads the bases, their packed copy, the table and its prefix directory
 * where `owner` has them (no byte copied, nothing computed; `owner` must outlive `ctx`); on another device the bases and the table are copied over the link
 * and the rest is made from them.  The host copy of isaac_gpu_set_host_contigs goes along.  A further worker or loader context of a device costs
 * milliseconds and no memory this way, against a 3 GB upload and 5.5 GB of directory and packed bases with isaac_gpu_load_contigs + isaac_gpu_share_index. */
int isaac_gpu_share_reference(isaac_gpu_ctx *ctx, isaac_gpu_ctx *owner);

/* sorted-reference.xml: reference::SortedReferenceMetadata::Contig / ::MaskFile (include/reference/SortedReferenceMetadata.hh:44-98) as
 * plain records.  strings are NUL-terminated. */
typedef struct
{
    uint64_t genomic_position, offset, size, total_bases, acgt_bases;
    uint32_t index, karyotype_index;
    char name[256], file[1024], bam_sq_as[256], bam_sq_ur[1024], bam_m5[64];
} isaac_reference_contig;
typedef struct { uint64_t kmers; uint32_t mask_width, mask, seed_length, reserved; char file[1024]; } isaac_reference_mask_file;

/* reference::loadSortedReferenceXml / saveSortedReferenceXml (lib/reference/SortedReferenceXml.cpp:193-213,216-330) on memory buffers;
 * no GPU involved.  parse: contigs / masks may be NULL to count; *format_version receives the version the metadata carries after
 * loading (the current one, as the reference bumps it).  format: xml_out may be NULL to size it; the text ends with a NUL that
 * *n_bytes_out does not count.  Errors: ISAAC_GPU_EFORMAT with the reader's message in isaac_gpu_sorted_reference_last_error(). */
int isaac_gpu_sorted_reference_parse(const char *xml_text, uint64_t n_bytes, isaac_reference_contig *contigs, uint32_t contig_capacity, uint32_t *n_contigs,
                                     isaac_reference_mask_file *masks, uint32_t mask_capacity, uint32_t *n_masks, uint32_t *format_version);
int isaac_gpu_sorted_reference_format(const isaac_reference_contig *contigs, uint32_t n_contigs, const isaac_reference_mask_file *masks, uint32_t n_masks,
                                      char *xml_out, uint64_t capacity, uint64_t *n_bytes_out);
const char *isaac_gpu_sorted_reference_last_error(void);
/* isaac-align -r <sorted-reference.xml>: maps the 32-mer mask files the XML names (relative paths: relative to the XML) and loads
 * them with the contig translation of <Index> / <KaryotypeIndex> (isaac_gpu_load_index).  The contigs must be loaded already, in
 * karyotype order. */
int isaac_gpu_load_sorted_reference(isaac_gpu_ctx *ctx, const char *xml_path);
/* isaac-sort-reference's output for the resident table (64 masks): <directory>/<genome_name>-32mer-6bit-ABCD-NN.dat and
 * <directory>/sorted-reference.xml with the given contig metadata */
int isaac_gpu_save_sorted_reference(isaac_gpu_ctx *ctx, const char *directory, const char *genome_name, const isaac_reference_contig *contigs, uint32_t n_contigs);

/* Replaces one tile's worth of alignWorkflow::FindMatchesTransition::findLaneMatches (both seed iterations;
 * lib/workflow/alignWorkflow/FindMatchesTransition.cpp:391-427): alignment::ClusterSeedGenerator::generateThread
 * (lib/alignment/ClusterSeedGenerator.cpp:138-192) + MatchFinder<KmerT>::findMatches (include/alignment/MatchFinder.hh:100-104).
 *   bcl_dev              n_clusters x (read_length[0] + read_length[1]) BCL bytes (base | quality << 2, 0 = N)
 *   matches_dev          receives what io::TileMatchWriter::write(SeedId, ReferencePosition) would have been called with
 *                        (include/io/MatchWriter.hh:72; the <Temp>/..._matches.dat files of lib/io/MatchWriter.cpp:79-94),
 *                        grouped by cluster; the order inside a cluster is unspecified, as in the reference's files
 *                        (thread-interleaved appends; SelectMatchesTransition.cpp:242-254 sorts them later).
 *   cluster_offsets_dev  n_clusters + 1 entries: cluster c owns matches_dev[offsets[c] .. offsets[c + 1]).  An empty range is
 *                        a cluster for which the reference stores only NoMatch records.
 *   contig_has_matches_host  n_contigs bytes, OR-ed: the "MatchDistribution::isEmptyContig" fact (MatchDistribution.hh:96-101)
 * The two buffers are the hand-over between the two halves of the path, exactly like the match files of the reference. */
int isaac_gpu_find_matches(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                           isaac_match *matches_dev, uint64_t capacity, uint64_t *cluster_offsets_dev, uint64_t *n_matches_out,
                           uint8_t *contig_has_matches_host);

/* Tells the extend stage which contigs the reference would have loaded (MatchSelector.cpp:85-90,138): the OR of
 * contig_has_matches over the whole run (all devices).  Lengths of the others count as 0 in the rest-of-genome
 * correction (include/alignment/RestOfGenomeCorrection.hh:44-55). NULL = all loaded. */
int isaac_gpu_set_loaded_contigs(isaac_gpu_ctx *ctx, const uint8_t *contig_loaded_host, uint32_t n_contigs);

/* Replaces alignment::FragmentBuilder::build for every cluster of the tile (include/alignment/FragmentBuilder.hh:62-70;
 * getFragments()/getCigarBuffer() :71-72): candidates in (cluster, read, list order); with_gaps and trim select the two ways
 * MatchSelector calls it (MatchSelector.cpp:233-245 vs :298-312).  candidates_dev / cigar_dev may be NULL. */
int isaac_gpu_build_fragments(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                              const isaac_match *matches_dev, const uint64_t *cluster_offsets_dev, int with_gaps, int trim,
                              isaac_candidate *candidates_dev, uint64_t capacity, uint64_t *n_candidates_out,
                              uint32_t *cigar_dev, uint64_t cigar_capacity, uint64_t *n_cigar_out);

/* Replaces MatchSelector::determineTemplateLength (lib/alignment/MatchSelector.cpp:188-256): learns the template length
 * statistics from the clusters of the tile, in cluster order, until they are stable. */
int isaac_gpu_determine_tls(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                            const isaac_match *matches_dev, const uint64_t *cluster_offsets_dev, isaac_tls *tls_out);

/* Replaces MatchSelector::processMatchList for the tile (lib/alignment/MatchSelector.cpp:258-368): fragment building with
 * gaps, TemplateBuilder::buildTemplate (shadow rescue, alignment scores), the semialigned / overlapping end clippers, and
 * the io::FragmentHeader fields FragmentCollector would store.  One record per read, in cluster order:
 * fragments_dev[cluster * n_reads + read]; its CIGAR is cigar_dev[cigar_offset .. + cigar_length);
 * cigar_dev must hold n_clusters * n_reads * ISAAC_GPU_MAX_CIGAR_OPS words.
 * The call returns when the records are complete, unless isaac_gpu_set_deferred_completion(ctx, 1) was called: see there.
 * isaac_fragment::reserved: bit 2 = a fixed internal capacity was exceeded for this cluster (result not exact; counted in
 * isaac_counters::overflow_clusters), bit 1 = the reference would not have stored the template (only without --keep-unaligned),
 * bit 3 = one of the cluster's alignment scores is unsigned(floor(-10 * log10(x))) (TemplateBuilder.cpp:273,437,604-608,912-920) with the
 * bits 16-31 = BamTemplate::getAlignmentScore as 16 bits (0xffff: unknown), which the duplicate ranking of the BAM stage needs;
 * [bit 3, continued] argument of floor within 1e-11 of an integer: the device's log10 / exp agree with glibc's to the last ulp or so, which can move the
 * floor only there.  A host that has to be certain re-derives exactly these clusters (counted in isaac_counters::mapq_near_integer;
 * a handful per million pairs). */
int isaac_gpu_select(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                     const isaac_match *matches_dev, const uint64_t *cluster_offsets_dev, const isaac_tls *tls,
                     isaac_fragment *fragments_dev, uint32_t *cigar_dev, uint64_t cigar_capacity);
/* The clusters of a selected tile whose MAPQ arithmetic came within 1e-11 of an integer on the device (isaac_fragment::reserved bit 3; isaac_counters::
 * mapq_near_integer: a handful per million pairs) redone on the host with glibc's exp / log10 / floor, which is what the reference computes with
 * (lib/alignment/TemplateBuilder.cpp:270-273,433-439): every such cluster goes through the thread-serial form of the path -- FragmentBuilder::build from its
 * seed matches, TemplateBuilder::buildTemplate with its mate rescues, clippers, records -- and where a record or CIGAR differs from the device's, the device's
 * is replaced in fragments_dev / cigar_dev.  To be called after isaac_gpu_select[_n] with the same arguments, before isaac_gpu_compact_cigars (cigar_dev in
 * slots of ISAAC_GPU_MAX_CIGAR_OPS words).  *n_flagged_out: clusters looked at; *n_changed_out: clusters whose records were replaced (none has ever been
 * seen to differ; a host that must be certain calls this).  The first call fetches the contigs back into host memory. */
/* bases_host: the contigs as isaac_gpu_load_contigs was given them, which the caller keeps for as long as the context lives (NULL: forget them); without it the
 * first isaac_gpu_resolve_flagged call fetches the contigs back from the device. */
int isaac_gpu_set_host_contigs(isaac_gpu_ctx *ctx, const char *bases_host);
int isaac_gpu_resolve_flagged(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile, const isaac_match *matches_dev, const uint64_t *cluster_offsets_dev,
                              const isaac_tls *tls, isaac_fragment *fragments_dev, uint32_t *cigar_dev, uint64_t *n_flagged_out, uint64_t *n_changed_out);
/* The same when the caller knows the tile's match count (*n_matches_out of isaac_gpu_find_matches): isaac_gpu_select reads it from
 * cluster_offsets_dev, which is a host wait for everything queued on the context's stream; with deferred completion this form queues its
 * work without one.  The count sizes the call's candidate pool (one slot per match): if it is smaller than the number of matches under
 * cluster_offsets_dev, the clusters beyond it get no candidates, are flagged (isaac_fragment::reserved bit 2, overflow_clusters) and the
 * call -- or, with deferred completion, the next isaac_gpu_synchronize -- returns ISAAC_GPU_ECAPACITY.  (Round 3 kept these counts in a
 * process-wide table keyed by the offsets pointer, which a recycled device address could make stale.) */
int isaac_gpu_select_n(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                       const isaac_match *matches_dev, uint64_t n_matches, const uint64_t *cluster_offsets_dev, const isaac_tls *tls,
                       isaac_fragment *fragments_dev, uint32_t *cigar_dev, uint64_t cigar_capacity);

/* The same from explicit candidate lists instead of match lists: TemplateBuilder::buildTemplate(contigList, restOfGenomeCorrection,
 * readMetadataList, sequencingAdapters, fragments, cluster, templateLengthStatistics) (include/alignment/TemplateBuilder.hh, the
 * overload the reference's own unit tests drive, lib/alignment/cppunit/testTemplateBuilder.cpp:149-373) followed by the clippers
 * and the record conversion of isaac_gpu_select.  candidates_dev: as isaac_gpu_build_fragments writes them, cluster by cluster,
 * read 0's list before read 1's, each in list order; candidate_offsets_dev[k] .. [k + 1]: the candidates of cluster k
 * (n_clusters + 1 entries); candidate_cigars_dev: the words isaac_candidate::cigar_offset / cigar_length refer to.
 * isaac_gpu_build_fragments(with_gaps = 1, trim = 1) followed by this call is isaac_gpu_select. */
int isaac_gpu_select_candidates(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, uint32_t n_clusters, uint32_t tile,
                                const isaac_candidate *candidates_dev, const uint64_t *candidate_offsets_dev, const uint32_t *candidate_cigars_dev,
                                const isaac_tls *tls, isaac_fragment *fragments_dev, uint32_t *cigar_dev, uint64_t cigar_capacity);

/* Packs the CIGARs of isaac_gpu_select's fixed 40-word slots back to back, in record order, and rewrites cigar_offset of every
 * record accordingly: the form in which io::FragmentHeader records are followed by their CIGAR in the reference's bin files
 * (include/io/Fragment.hh:73-100) and in which a tile's result crosses PCIe or xGMI (about 2 words per read instead of 40).
 * cigar_out_dev must not overlap cigar_in_dev; *n_words_out receives the packed length (also when it exceeds capacity: the call then
 * returns ISAAC_GPU_ECAPACITY and has changed nothing, the records still refer to their slots, so it can be repeated with a larger pool).
 * isaac_gpu_compact_cigars_async is the same without the host wait: the packed length is written to the DEVICE word *n_words_out_dev
 * behind the kernels on the context's stream, and a pool that is too small shows as *n_words_out_dev > capacity with nothing changed. */
int isaac_gpu_compact_cigars(isaac_gpu_ctx *ctx, isaac_fragment *fragments_dev, uint64_t n_records, const uint32_t *cigar_in_dev,
                             uint32_t *cigar_out_dev, uint64_t capacity, uint64_t *n_words_out);
int isaac_gpu_compact_cigars_async(isaac_gpu_ctx *ctx, isaac_fragment *fragments_dev, uint64_t n_records, const uint32_t *cigar_in_dev,
                                   uint32_t *cigar_out_dev, uint64_t capacity, uint64_t *n_words_out_dev);

/* The output side of the path: the BAM alignment records build::Build writes
 * (lib/build/Build.cpp, lib/build/BinSorter.cpp) from what isaac_gpu_select produced, computed where the records already are.
 *   order    PackedFragmentBuffer::orderForBam (include/build/PackedFragmentBuffer.hh:149-176): bin position, global cluster id
 *            (tile * 1000000000 + cluster, include/build/FragmentIndex.hh:33), mapped before unmapped (a shadow follows its
 *            singleton), first read before second; templates with both reads unaligned last, in (tile, cluster, read) order
 *            (--keep-unaligned back); records flagged "not stored" (isaac_fragment::reserved bit 1) are left out
 *   realignment BinSorter::collectGaps / realignGaps (lib/build/BinSorter.cpp:387-417) with build::GapRealigner (lib/build/GapRealigner.cpp) and
 *            build::SemialignedEndsClipper when isaac_bam_options::realign_gaps is set: every fragment is tried against the gaps the other
 *            fragments of its contig carry (mismatch 3, gap open 4, gap extend 0; at most ten gaps in reach), positions, CIGARs, edit
 *            distances, TLEN, mate positions and proper-pair flags change on a private copy of the records before they are ordered;
 *            every contig is one bin (the reference's bins, whose ends limit what it realigns, depend on its memory settings)
 *   duplicates  BinSorter::resolveDuplicates (lib/build/BinSorter.cpp:293-330) with DuplicatePairEndFilter over FDuplicateFilter / RSDuplicateFilter
 *            (include/build/DuplicatePairEndFilter.hh, DuplicateFragmentIndexFiltering.hh) when isaac_bam_options asks for it: one library
 *            (--single-library-samples with one barcode), all the tiles of the call compared with each other as if every contig were one
 *            bin of the reference (its result depends on where its bins end); the template's rank of io::getTemplateDuplicateRank is
 *            derived from the BCL qualities, the two records and isaac_fragment::reserved bits 16-31 (the template's alignment score)
 *   record   bam::serializeAlignment over build::FragmentAccessorBamAdapter (include/bam/Bam.hh:257-345,
 *            include/build/FragmentAccessorBamAdapter.hh:127-377) with the default tag set SM AS RG NM BC OC (--bam-exclude-tags ZX,ZY; OC, the CIGAR before
 *            realignment, only on records the gap realigner changed);
 *            bases and qualities as FragmentCollector::storeBclAndCigar keeps them (lib/alignment/matchSelector/FragmentCollector.cpp:84-111)
 * One tile = the buffers of one isaac_gpu_select call (bcl_dev as given to it, its records and cigar pool, fixed slots or packed);
 * read_name_prefix = "<flowcell id>:<lane>:<tile>:" (FragmentAccessorBamAdapter::readName), at most 63 characters.  Contig ids are
 * written as they are (the BAM header must list the contigs in the order of isaac_gpu_load_contigs).
 * bam_dev receives the uncompressed records back to back; *n_bytes_out their length (also when it exceeds capacity),
 * *n_records_out their number, *unaligned_offset_out the offset of the first record of the unaligned bin (= *n_bytes_out if none). */
typedef struct
{
    const uint8_t *bcl_dev; const isaac_fragment *fragments_dev; const uint32_t *cigar_dev; uint64_t n_records; const char *read_name_prefix;
    const char *read_group;   /* RG:Z of this tile's records when lanes differ (one 'none' barcode per lane without a sample sheet,
                                 lib/demultiplexing/SampleSheetCsv.cpp:101-112); NULL = isaac_bam_options::read_group; at most 27 characters */
    const isaac_tls *tls;     /* the template length statistics this tile's isaac_gpu_select ran with, when they differ between tiles (the reference
                                 keeps them per barcode, that is per lane: MatchSelector.cpp:395-412); NULL = isaac_bam_options::tls */
} isaac_bam_tile;
typedef struct
{
    uint32_t forced_dodgy_alignment_score;  /* MAPQ of alignments whose score is unknown (0xffff): --dodgy-alignment-score */
    uint32_t pessimistic_mapq;               /* --pessimistic-mapq: min instead of max of SM and AS for proper pairs */
    const char *read_group;                  /* RG:Z value: the barcode index (FragmentAccessorBamAdapter.hh:283-299); NULL = "0" */
    const char *barcode;                     /* BC:Z value: the sample sheet barcode name (:307-335); NULL = "none" */
    uint32_t mark_duplicates;                /* --mark-duplicates (reference default 1): duplicates get BAM flag 0x400 */
    uint32_t keep_duplicates;                /* --keep-duplicates (reference default 1): 0 leaves duplicates out of the file */
    uint32_t realign_gaps;                   /* --realign-gaps: 0 = no, 1 = sample / project / all (one gap group: the call's records are one sample) */
    uint32_t realign_vigorously;             /* --realign-vigorously (reference default 0): a realigned fragment is tried again until nothing improves, and fragments with more than ten (up to thirty) gaps in reach are tried too (GapRealigner.cpp:1117,1241) */
    uint32_t realign_dodgy;                  /* --realign-dodgy (reference default 0) */
    const isaac_tls *tls;                    /* the template length statistics isaac_gpu_select ran with: GapRealigner::updatePairDetails re-derives the
                                                proper-pair flag of realigned pairs from them; required with realign_gaps and paired reads
                                                unless every tile brings its own */
    /* One bin of the file at a time (build::Build works through its bins one by one, lib/build/Build.cpp:509-543,793-900): with bin_filter
     * set the call writes only the records whose bin position lies on contigs [bin_first_contig, bin_end_contig) -- plus, with bin_unaligned,
     * the templates without a position -- while every record of the tiles still serves as its mate's mate (duplicate ranks, pair details
     * after realignment).  The tiles of such a call are what isaac_gpu_bin_tile made for the bin.  Bins of whole contigs written in contig
     * order, the unaligned one last, give the very bytes of one call over all tiles. */
    uint32_t bin_filter /* 0: no, 1: by contigs, 2: by positions (below) */, bin_first_contig, bin_end_contig, bin_unaligned;
    /* NULL, or room for one entry per record of the call's tiles: the call leaves there, in file order, what the BAM index wants to know about
     * every record it writes (isaac_gpu_bam_indexer_add_entries), so that a host need not fetch and parse the record stream itself. */
    struct isaac_bam_index_entry *index_entries_dev;
    /* bin_filter 2: the bin is the positions [bin_first_position, bin_end_position), both in the encoding of isaac_fragment::f_strand_position
     * (reference::ReferencePosition values, which order by contig, then position): a run of small contigs, or a stretch of a large one as
     * alignment::BinMetadata describes it (the reference cuts its contigs into bins of --target-bin-size, include/alignment/matchSelector/BinIndexMap.hh:44-104).
     * A bin is sorted, filtered for duplicates and realigned by itself (lib/build/BinSorter.cpp:293-330,387-417): only the bin's own records
     * take part, gaps are looked for between its first and last position, a pair whose mate lies in another bin is not realigned. */
    uint64_t bin_first_position, bin_end_position;
} isaac_bam_options;
/* offset and length of the record in the call's stream, refID, pos, FLAG and l_seq as written, the reference bases its CIGAR covers */
typedef struct isaac_bam_index_entry { uint64_t offset; uint32_t bytes; int32_t ref_id; int32_t pos; uint32_t flag, seq_length, observed; } isaac_bam_index_entry;
int isaac_gpu_bam_records(isaac_gpu_ctx *ctx, const isaac_bam_tile *tiles, uint32_t n_tiles, const isaac_bam_options *options /* NULL = defaults */,
                          uint8_t *bam_dev, uint64_t capacity, uint64_t *n_bytes_out, uint64_t *n_records_out, uint64_t *unaligned_offset_out);

/* Replaces alignment::matchSelector::BinningFragmentStorage (lib/alignment/matchSelector/BinningFragmentStorage.cpp, FragmentBinner.cpp): while
 * the reference selects matches it writes every fragment -- header, bases, CIGAR -- to the file of the bin its position falls into, and builds the
 * BAM bin by bin from those files, so that a run never has to fit memory.  Here: the output of one isaac_gpu_select call (cigar_dev packed by
 * isaac_gpu_compact_cigars or not) is cut into one compact tile per bin, each holding the clusters with at least one stored record in the bin:
 * their BCL bytes, their records (n_reads per cluster, in cluster order; cigar_offset relative to the tile's own words) and their CIGAR words.
 * A pair whose reads lie in two bins goes to both, whole.  bin_of_contig[c]: the bin of contig c (< n_bins - 1; bins of whole contigs in contig
 * order; isaac_gpu_bin_tile_map: bins that are stretches of a contig as well); bin n_bins - 1 takes the templates without a position.  out_dev receives, bin after bin, each part starting on a multiple of 64 bytes:
 *     n_clusters x cluster length bytes of BCL | n_clusters x n_reads records | n_cigar_words words
 * sizes_out[b]: the two counts of bin b; *n_bytes_out: the bytes written (or needed, with ISAAC_GPU_ECAPACITY).  A part's three arrays, copied
 * to wherever the bin is kept and back to a device, are the bcl_dev / fragments_dev / cigar_dev of an isaac_bam_tile with the original tile's name
 * prefix, read group and statistics. */
typedef struct { uint64_t n_clusters, n_cigar_words; } isaac_bin_size;
/* The bins of a run in file order (alignment::matchSelector::BinIndexMap, include/alignment/matchSelector/BinIndexMap.hh:44-104: every contig starts a
 * bin of its own there and goes on into further bins by the match distribution; here the caller decides): bin_of_contig[c] is the first bin of contig c
 * -- consecutive small contigs may share one --, and a contig goes on into the next bin at each of its cut_positions (ascending values in the encoding
 * of isaac_fragment::f_strand_position).  Up to 65535 bins; the last one takes the templates without a position. */
typedef struct { const uint32_t *bin_of_contig; uint32_t n_contigs; const uint64_t *cut_positions; uint32_t n_cuts; uint32_t n_bins; } isaac_bin_map;
int isaac_gpu_bin_tile_map(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, const isaac_fragment *fragments_dev, const uint32_t *cigar_dev, uint32_t n_clusters,
                           const isaac_bin_map *map, uint8_t *out_dev, uint64_t capacity, isaac_bin_size *sizes_out, uint64_t *n_bytes_out);
int isaac_gpu_bin_tile(isaac_gpu_ctx *ctx, const uint8_t *bcl_dev, const isaac_fragment *fragments_dev, const uint32_t *cigar_dev, uint32_t n_clusters,
                       const uint32_t *bin_of_contig, uint32_t n_contigs, uint32_t n_bins, uint8_t *out_dev, uint64_t capacity, isaac_bin_size *sizes_out, uint64_t *n_bytes_out);

/* Host-only pieces of the BAM file (no context, no GPU).  Errors: isaac_gpu_bam_last_error().
 * isaac_gpu_bam_header: bam::serializeHeader (include/bam/Bam.hh:153-235): magic, the text (@HD VN:1.0 SO:coordinate, @PG ID:iSAAC
 * with CL / DS / VN, header_lines such as --bam-header-tag and the @RG lines verbatim, one @SQ SN LN [AS] [UR] [M5] per contig -- the three
 * optional tags from isaac_reference_contig::bam_sq_as / bam_sq_ur / bam_m5, arrays or entries may be NULL or empty --), the contig table.
 * isaac_gpu_bgzf_compress: the BGZF framing of bgzf::BgzfCompressor (include/bgzf/BgzfCompressor.hh:36-176): blocks of at most
 * 0xFFFF - 41 input bytes, each a gzip member with the BC extra field, deflated with zlib at `level` (--bam-gzip-level) on
 * n_threads threads; eof_block != 0 appends the 28-byte empty block of bam::serializeBgzfFooter (lib/bam/Bam.cpp:38-45).
 * out_host must hold isaac_gpu_bgzf_bound(n_bytes) bytes. */
const char *isaac_gpu_bam_last_error(void);
int isaac_gpu_bam_header(const char *command_line, const char *description, const char *version, const char *const *header_lines, uint32_t n_header_lines,
                         const char *const *contig_names, const uint32_t *contig_lengths, const char *const *contig_as, const char *const *contig_ur,
                         const char *const *contig_m5, uint32_t n_contigs, uint8_t *out_host, uint64_t capacity, uint64_t *n_bytes_out);
uint64_t isaac_gpu_bgzf_bound(uint64_t n_bytes);
int isaac_gpu_bgzf_compress(const uint8_t *data_host, uint64_t n_bytes, int level, uint32_t n_threads, int eof_block,
                            uint8_t *out_host, uint64_t capacity, uint64_t *n_bytes_out);

/* sorted.bam.bai: bam::BamIndexPart / bam::BamIndex (lib/bam/BamIndexer.cpp:43-126,129-472; constants include/bam/BamIndexer.hh:44-55,646-647), host-only.
 * The reference indexes its output bin by bin: a bin's records give chunks and 16 kb linear-index entries in offsets of the bin's own
 * uncompressed bytes, which the BGZF blocks the bin was compressed to turn into virtual file offsets.  A part is that: records_bytes bytes
 * of the uncompressed record stream at records_offset which were compressed on their own into the bgzf_bytes bytes at bgzf_host (whole
 * BGZF blocks), the parts following each other in the file behind header_bgzf_bytes bytes of compressed header.  Parts with aligned
 * records must come in contig order (a contig may have several); the records of the unaligned bin are a part of their own.
 * bai_out may be NULL with capacity 0 to size the file.  Errors: isaac_gpu_bam_index_last_error(). */
typedef struct { uint64_t records_offset, records_bytes; const uint8_t *bgzf_host; uint64_t bgzf_bytes; } isaac_bam_index_part;
const char *isaac_gpu_bam_index_last_error(void);
int isaac_gpu_bam_index(const uint8_t *records_host, const isaac_bam_index_part *parts, uint32_t n_parts, uint32_t n_contigs, uint64_t header_bgzf_bytes,
                        uint8_t *bai_out, uint64_t capacity, uint64_t *n_bytes_out);
/* The same index made as the reference makes it: one bin at a time while the file is written (bam::BamIndex::processIndexPart per saved bin,
 * lib/build/Build.cpp), so that the record stream of a whole run never has to be in one place.  add: the part's uncompressed records and the BGZF
 * blocks they were compressed to (neither is kept); parts in file order, as above.  finish may be called once all parts are in. */
typedef struct isaac_bam_indexer isaac_bam_indexer;
isaac_bam_indexer *isaac_gpu_bam_indexer_create(uint32_t n_contigs, uint64_t header_bgzf_bytes);
int isaac_gpu_bam_indexer_add(isaac_bam_indexer *indexer, const uint8_t *records_host, uint64_t records_bytes, const uint8_t *bgzf_host, uint64_t bgzf_bytes);
int isaac_gpu_bam_indexer_add_entries(isaac_bam_indexer *indexer, const isaac_bam_index_entry *entries_host, uint64_t n_entries, uint64_t records_bytes, const uint8_t *bgzf_host, uint64_t bgzf_bytes);
int isaac_gpu_bam_indexer_finish(isaac_bam_indexer *indexer, uint8_t *bai_out, uint64_t capacity, uint64_t *n_bytes_out);
void isaac_gpu_bam_indexer_destroy(isaac_bam_indexer *indexer);

/* BGZF framing without compression on the device, for --bam-gzip-level 0: what bgzf::BgzfCompressor produces at gzip level 0
 * (include/bgzf/BgzfCompressor.hh:36-176: blocks of at most 0xFFFF - 41 input bytes, each a gzip member with the BC extra field around
 * one stored deflate block, CRC-32 and length behind it), byte for byte what isaac_gpu_bgzf_compress(level 0) writes.  data_dev: n_bytes
 * of an uncompressed BAM stream in HBM (header or records); out_dev must hold isaac_gpu_bgzf_store_bound(n_bytes) bytes; eof_block != 0
 * appends the 28-byte empty block.  The CRC-32 of every block is computed on the device. */
uint64_t isaac_gpu_bgzf_store_bound(uint64_t n_bytes);
int isaac_gpu_bgzf_store(isaac_gpu_ctx *ctx, const uint8_t *data_dev, uint64_t n_bytes, int eof_block, uint8_t *out_dev, uint64_t capacity, uint64_t *n_bytes_out);

/* BGZF with compression on the device, for --bam-gzip-level 1 and up: replaces bgzf::BgzfCompressor with zlib behind it
 * (include/bgzf/BgzfCompressor.hh:36-176, wired into the BAM writer at lib/build/Build.cpp:181-254) for streams that are in HBM already.  Same
 * framing as above (gzip members with the BC field, CRC-32 and length) around blocks of 20 KB of input -- smaller than the reference's
 * 0xFFFF - 41, which is as valid a BGZF file and lets a compute unit work on five blocks at a time; inside, one
 * dynamic-Huffman deflate block per member (hash-table LZ77 over the block, the call's two Huffman tables made from a sample of its blocks),
 * or a stored block where that is not smaller.  The compressed bytes are not zlib's: what is identical is what they inflate to.  out_dev:
 * isaac_gpu_bgzf_deflate_bound(n_bytes) bytes are always enough; with less the call fails with ISAAC_GPU_ECAPACITY once the output does not
 * fit (*n_bytes_out then holds the bound). */
uint64_t isaac_gpu_bgzf_deflate_bound(uint64_t n_bytes);
int isaac_gpu_bgzf_deflate(isaac_gpu_ctx *ctx, const uint8_t *data_dev, uint64_t n_bytes, int eof_block, uint8_t *out_dev, uint64_t capacity, uint64_t *n_bytes_out);

/* The leaf: alignment::BandedSmithWaterman::align (include/alignment/BandedSmithWaterman.hh:75-86) for a batch;
 * scores as the reference constructor takes them (GappedAligner.cpp:41-42: match, mismatch, -gapOpen, -gapExtend).
 * results[i].n_ops == 0xffffffff flags a CIGAR longer than ISAAC_GPU_MAX_CIGAR_OPS. */
int isaac_gpu_bsw_batch(isaac_gpu_ctx *ctx, int match, int mismatch, int gap_open, int gap_extend,
                        const char *sequences_dev, const isaac_bsw_job *jobs_dev, uint32_t n_jobs, uint32_t max_query_length,
                        isaac_bsw_result *results_dev);

/* The data format on the input side of the path: io::FastqReader::next / extractBcl (lib/io/FastqReader.cpp:120-283,
 * include/io/FastqReader.hh:144-210) and io::FastqLoader::loadSingleRead (include/io/FastqLoader.hh) for one read of a lane.
 * fastq_dev: n_bytes of uncompressed FASTQ text resident in HBM (gzip inflation stays with the caller).  The records are
 * parsed with the reference's tolerance (any run of \r / \n separates lines, the '+' line may repeat the header, a sequence
 * line that starts with '+' is a zero-length read) and converted to BCL bytes (base | quality << 2, 0 for N), cluster k at
 * bcl_dev[k * cluster_length + offset of read read_index], at most max_clusters of them.
 * final = 0: the text is a piece of a longer file; a record not yet followed by a newline is left for the next call and
 * *consumed_bytes_out is where that call's text has to start.  final = 1: the text ends at the end of the file.
 * Errors as the reference throws them, first in file order: ISAAC_GPU_EFORMAT (bad structure, base or quality; quality
 * must be in [0, 63]) or ISAAC_GPU_EREADLEN (record shorter than the read length unless allow_variable_length, which pads
 * with N); *n_clusters_out then counts the records before the bad one and *error_offset_out is the byte offset the
 * reference would report.  Reads longer than the read length are cut.  Non-contiguous cycle lists (use-bases masks) are
 * not supported. */
int isaac_gpu_fastq_to_bcl(isaac_gpu_ctx *ctx, const char *fastq_dev, uint64_t n_bytes, uint32_t read_index,
                           int allow_variable_length, int final, uint8_t *bcl_dev, uint32_t max_clusters,
                           uint32_t *n_clusters_out, uint64_t *consumed_bytes_out, uint64_t *error_offset_out);

/* How the clusters of one load of a lane's FASTQ files become tiles (host-only, no context): FastqSeedSource's tileClustersMax_
 * (lib/workflow/alignWorkflow/FastqDataSource.cpp:82-84: min(--clusters-at-a-time, 40000000 / #seeds), or the latter alone when the
 * option is 0) and the tile breakdown of discoverTiles (:153-173).  first_tile is 1 for a lane's first load and *next_tile_out of the
 * previous load afterwards; cluster ids restart at 0 in every tile (the tile number and the id inside it are what isaac_gpu_find_matches /
 * isaac_gpu_select take and what the BAM read name shows).  Returns ISAAC_GPU_ECAPACITY (with *n_tiles_out set) when the arrays are short. */
uint32_t isaac_gpu_fastq_tile_clusters_max(uint32_t clusters_at_a_time, uint32_t n_seeds);
int isaac_gpu_fastq_tiles(uint32_t clusters_loaded, uint32_t clusters_at_a_time, uint32_t n_seeds, uint32_t first_tile,
                          uint32_t *tile_numbers, uint32_t *tile_clusters, uint32_t capacity, uint32_t *n_tiles_out, uint32_t *next_tile_out);

/* The option defaults a host starts from (host-only): options::AlignOptions (lib/options/AlignOptions.cpp:77-160) for reads of the given
 * lengths (read_length2 = 0: single-ended) with --gap-scoring bwa, --seeds auto and the first-pass rule of :1165-1171.
 * isaac_gpu_parse_gap_scoring: --gap-scoring "bwa" | "eland" | m:mm:go:ge:mge (AlignOptions.cpp:689-743).
 * isaac_gpu_parse_seeds: --seeds "auto" | "all" | offsets "0:32:64[,0:32]" per read (lib/options/alignOptions/SeedDescriptorOption.cpp:38-244)
 * for params->n_reads / read_length / seed_length, with --first-pass-seeds as given (it becomes 2 with auto seeds and a non-zero
 * semialigned_gap_limit, and never exceeds what a read has room for).  ISAAC_GPU_EINVAL with the reference's message in
 * isaac_gpu_params_last_error(). */
const char *isaac_gpu_params_last_error(void);
int isaac_gpu_default_params(uint32_t read_length1, uint32_t read_length2, isaac_params *out);
int isaac_gpu_parse_gap_scoring(const char *gap_scoring, isaac_params *params);
int isaac_gpu_parse_seeds(const char *descriptor, uint32_t first_pass_seeds, isaac_params *params);
/* options::parseDefaultAdapters (lib/options/alignOptions/DefaultAdaptersOption.cpp:35-60) with flowcell::SequencingAdapterListGrammar
 * (include/flowcell/SequencingAdapterListGrammar.hpp:52-104): one --default-adapters entry -- "Standard", "Nextera", "NexteraMp"
 * (lib/flowcell/SequencingAdapterMetadata.cpp:29-39) or a comma-separated list of ACGT / ACGT* / *ACGT -- into params->adapters.
 * ISAAC_GPU_EINVAL with "Could not parse the default-adapters ..." as isaac_gpu_last_error() for anything else. */
int isaac_gpu_parse_adapters(const char *descriptor, isaac_params *params);

int isaac_gpu_get_counters(isaac_gpu_ctx *ctx, isaac_counters *out);
/* average device time (ms) of the named launch sequence over the launches since the last reset, measured with HIP events on the
 * stream it runs on; names: "find_matches", "compact_matches", "build_fragments", "align_candidates", "finish_candidates",
 * "indel_fragments", "gapped_fragments" (k_gapped_jobs of the fragment stage) and "gapped_fragments_rescan", "finish_fragments", "load_candidates", "plan_rescue", "rescue_windows", "rescue_align",
 * "rescue_gapped_plan", "gapped_rescue" / "gapped_rescue_rescan" (the same two kernels for the mate rescue), "sums_wave", "sums_large", "sums_xl", "sums_huge", "select", "select_heavy", "select_residual" (the last two
 * are launched for every chunk and read their cluster count on the device: a few microseconds when it is zero), "fastq_to_bcl", "bsw",
 * "bam_order", "bam_encode", "bgzf_store" */
int isaac_gpu_kernel_time_ms(isaac_gpu_ctx *ctx, const char *kernel, double *avg_ms, uint64_t *launches);
int isaac_gpu_reset_timers(isaac_gpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
