/* include/isaac_gpu.h is a C header: this file is compiled as C99 and takes the address of every entry point */
#include "isaac_gpu.h"
#include <stddef.h>

typedef void (*any_function)(void);
any_function isaac_gpu_entry_points[] = {
    (any_function)isaac_gpu_last_error, (any_function)isaac_gpu_create, (any_function)isaac_gpu_destroy, (any_function)isaac_gpu_malloc, (any_function)isaac_gpu_free,
    (any_function)isaac_gpu_upload, (any_function)isaac_gpu_download, (any_function)isaac_gpu_memory_info, (any_function)isaac_gpu_host_malloc, (any_function)isaac_gpu_host_free, (any_function)isaac_gpu_synchronize, (any_function)isaac_gpu_set_deferred_completion,
    (any_function)isaac_gpu_load_contigs, (any_function)isaac_gpu_load_contigs_dev, (any_function)isaac_gpu_load_index, (any_function)isaac_gpu_build_index,
    (any_function)isaac_gpu_get_index, (any_function)isaac_gpu_get_index_range, (any_function)isaac_gpu_get_mask_offsets,
    (any_function)isaac_gpu_sorted_reference_parse, (any_function)isaac_gpu_sorted_reference_format, (any_function)isaac_gpu_sorted_reference_last_error,
    (any_function)isaac_gpu_load_sorted_reference, (any_function)isaac_gpu_save_sorted_reference, (any_function)isaac_gpu_find_matches, (any_function)isaac_gpu_set_loaded_contigs,
    (any_function)isaac_gpu_build_fragments, (any_function)isaac_gpu_determine_tls, (any_function)isaac_gpu_select, (any_function)isaac_gpu_select_n, (any_function)isaac_gpu_select_candidates,
    (any_function)isaac_gpu_compact_cigars, (any_function)isaac_gpu_compact_cigars_async, (any_function)isaac_gpu_set_params, (any_function)isaac_gpu_index_dev, (any_function)isaac_gpu_set_index_dev, (any_function)isaac_gpu_bsw_batch, (any_function)isaac_gpu_fastq_to_bcl, (any_function)isaac_gpu_get_counters,
    (any_function)isaac_gpu_kernel_time_ms, (any_function)isaac_gpu_reset_timers,
    (any_function)isaac_gpu_bam_records, (any_function)isaac_gpu_bam_last_error, (any_function)isaac_gpu_bam_header, (any_function)isaac_gpu_bgzf_bound,
    (any_function)isaac_gpu_bgzf_compress, (any_function)isaac_gpu_fastq_tile_clusters_max, (any_function)isaac_gpu_fastq_tiles, (any_function)isaac_gpu_bgzf_store_bound, (any_function)isaac_gpu_bgzf_store, (any_function)isaac_gpu_share_index, (any_function)isaac_gpu_bin_tile, (any_function)isaac_gpu_bin_tile_map, (any_function)isaac_gpu_resolve_flagged, (any_function)isaac_gpu_set_host_contigs, (any_function)isaac_gpu_download_async, (any_function)isaac_gpu_download_wait, (any_function)isaac_gpu_share_reference, (any_function)isaac_gpu_bam_indexer_create, (any_function)isaac_gpu_bam_indexer_add, (any_function)isaac_gpu_bam_indexer_add_entries, (any_function)isaac_gpu_bam_indexer_finish, (any_function)isaac_gpu_bam_indexer_destroy, (any_function)isaac_gpu_bgzf_deflate_bound, (any_function)isaac_gpu_bgzf_deflate,
    (any_function)isaac_gpu_copy, (any_function)isaac_gpu_bam_index, (any_function)isaac_gpu_bam_index_last_error, (any_function)isaac_gpu_default_params,
    (any_function)isaac_gpu_parse_gap_scoring, (any_function)isaac_gpu_parse_seeds, (any_function)isaac_gpu_parse_adapters, (any_function)isaac_gpu_params_last_error,
};
size_t isaac_gpu_entry_point_count(void) { return sizeof(isaac_gpu_entry_points) / sizeof(isaac_gpu_entry_points[0]); }
int isaac_gpu_struct_sizes_ok(void)
{
    return sizeof(isaac_match) == 16 && sizeof(isaac_reference_kmer) == 16 && sizeof(isaac_fragment) == 64 && sizeof(isaac_candidate) == 96 && sizeof(isaac_bsw_job) == 24;
}
