// What the translation units of the library share: buffer descriptors handed to the kernels, launch constants, the work-counter
// reduction, and the prototypes of the kernels that live in kernels_*.hip (each kernel family is compiled on its own, so that a
// change to one does not rebuild the others and the build runs in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/isaac_gpu.h"
#include "cluster_ops.h"
#include "sums.h"

// occupancy target (waves per SIMD) of the thread-per-cluster kernels of the fragment stage; the register allocator spills to meet it
#ifndef ISAAC_FRAGMENT_WAVES
#define ISAAC_FRAGMENT_WAVES 6
#endif

using namespace isaac;

// run constants of the template kernels in device memory: passed by value they end up as private copies (dynamic indexing)
struct TemplateConstants { DevParams P; DevTls tls; RogCorrection rog; };

// ------------------------------------------------------------------------------------------------------------------
// wave-level reduction of the work counters, one atomic per field per wave.  The totals are kept in COUNTER_SHARDS copies
// (a block adds to the copy of its index; isaac_gpu_get_counters sums them): atomics on one address execute one after the
// other in L2, and a grid of small waves can spend longer queueing there than working.  Fields no lane touched cost a vote.
static const u32 COUNTER_SHARDS = 64;
__device__ inline void flushCounters(const Counters &local, Counters *global)
{
    const u64 *src = reinterpret_cast<const u64 *>(&local);
    u64 *dst = reinterpret_cast<u64 *>(global + (blockIdx.x & (COUNTER_SHARDS - 1)));
    for (u32 f = 0; f < sizeof(Counters) / sizeof(u64); ++f)
    {
        u64 v = src[f];
        if (!__any(v != 0)) continue;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(reinterpret_cast<unsigned long long *>(dst + f), static_cast<unsigned long long>(v));
    }
}

// one field, for the kernels that count one or two things (a Counters in registers is forty of them)
__device__ inline void flushCounter(u64 Counters::*field, u64 v, Counters *global)
{
    if (!__any(v != 0)) return;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(reinterpret_cast<unsigned long long *>(&(global[blockIdx.x & (COUNTER_SHARDS - 1)].*field)), static_cast<unsigned long long>(v));
}

// ------------------------------------------------------------------------------------------------------------------
// Quality::logMatchLookup / logMismatchLookup staged in LDS: every base of every alignment reads one of the two
#define ISAAC_STAGE_QUALITY_TABLES(R_IN, R_OUT)                                                                              \
    __shared__ double qualityTables[128];                                                                                    \
    for (u32 qi = threadIdx.x; qi < 128; qi += blockDim.x) qualityTables[qi] = qi < 64 ? (R_IN).logMatch[qi] : (R_IN).logMismatch[qi - 64]; \
    __syncthreads();                                                                                                         \
    DevReference R_OUT = (R_IN); R_OUT.logMatch = qualityTables; R_OUT.logMismatch = qualityTables + 64; R_OUT.logStride = 1;

// The same with a copy of the tables per lane of a half wavefront: entry q of copy c at [q * 32 + c].  A ds_read_b64 is served half a wavefront at
// a time, 64 banks of four bytes: lane l reading copy l % 32 always hits banks 2 (l % 32) and 2 (l % 32) + 1, whatever its q, so the 32 lanes of a
// half never meet (one shared table: a quarter to two fifths of the LDS cycles of the ungapped scans were conflicts, profiles/r3_zz_pmc_summary.json).
#ifndef ISAAC_QUALITY_TABLE_COPIES
#define ISAAC_QUALITY_TABLE_COPIES 32
#endif
#ifndef ISAAC_SCAN_TABLES_PER_LANE
#define ISAAC_SCAN_TABLES_PER_LANE 1
#endif
#define ISAAC_STAGE_QUALITY_TABLES_PER_LANE(R_IN, R_OUT)                                                                     \
    __shared__ double qualityTables[128 * ISAAC_QUALITY_TABLE_COPIES];                                                       \
    for (u32 qi = threadIdx.x; qi < 128 * ISAAC_QUALITY_TABLE_COPIES; qi += blockDim.x)                                      \
    { const u32 qe = qi / ISAAC_QUALITY_TABLE_COPIES; qualityTables[qi] = qe < 64 ? (R_IN).logMatch[qe] : (R_IN).logMismatch[qe - 64]; } \
    __syncthreads();                                                                                                         \
    DevReference R_OUT = (R_IN); R_OUT.logMatch = qualityTables + (threadIdx.x % ISAAC_QUALITY_TABLE_COPIES);                \
    R_OUT.logMismatch = R_OUT.logMatch + 64 * ISAAC_QUALITY_TABLE_COPIES; R_OUT.logStride = ISAAC_QUALITY_TABLE_COPIES;
// the ungapped scans of the fragment stage (k_align_candidates): a thread per alignment, two table reads per base
#if ISAAC_SCAN_TABLES_PER_LANE
#define ISAAC_STAGE_SCAN_TABLES(R_IN, R_OUT) ISAAC_STAGE_QUALITY_TABLES_PER_LANE(R_IN, R_OUT)
#else
#define ISAAC_STAGE_SCAN_TABLES(R_IN, R_OUT) ISAAC_STAGE_QUALITY_TABLES(R_IN, R_OUT)
#endif

// the chunk's gapped (banded Smith-Waterman) problems: written by the per-cluster threads, run by k_gapped_jobs
struct GappedBuffers { GappedJob *jobs; GappedResult *results; u32 cap; u32 *counter; u32 *base; };

// list lengths up to which the lean forms of the fragment stage's steps take a cluster (fragment_lean.h; at most LEAN_LIST_MAX = 16): their
// key area in LDS is sized by these, and with it the wavefronts a CU holds
#ifndef ISAAC_BUILD_LEAN_MAX
#define ISAAC_BUILD_LEAN_MAX 16
#endif
#ifndef ISAAC_FINISH_LEAN_MAX
#define ISAAC_FINISH_LEAN_MAX 8
#endif
static const u32 BUILD_LEAN_MAX = ISAAC_BUILD_LEAN_MAX, FINISH_LEAN_MAX = ISAAC_FINISH_LEAN_MAX;
static const u32 GENERAL_STAGE_MATCHES = 160;         // matches per cluster the general form of k_build_fragments keeps in LDS
static const u32 GENERAL_BLOCKS = 1024;               // workgroups of the general forms (each with its own banded-SW flag area for the serial fallback)

#ifndef ISAAC_BUILD_STAGE_MATCHES
#define ISAAC_BUILD_STAGE_MATCHES 16
#endif
static const u32 BUILD_STAGE_MATCHES = ISAAC_BUILD_STAGE_MATCHES;      // k_build_fragments: clusters with up to this many matches keep them in LDS

#ifndef ISAAC_SUMS16_GROUPS
#define ISAAC_SUMS16_GROUPS 8
#endif
static const u32 SUMS16_GROUPS = ISAAC_SUMS16_GROUPS;        // clusters (quarter wavefronts) per workgroup of k_cluster_sums16

// the chunk's ungapped alignment problems: (cluster << 8) | (read << 7) | index in the read's candidate list
struct AlignList { u32 *entries; u32 cap; u32 *counter; };

// ------------------------------------------------------------------------------------------------------------------
// Template stage.  Mate rescue (ShadowAligner::rescueShadow) is the bulk of the work of the select phase: a 7-mer scan of
// a window of several hundred reference bases plus one 150-base ungapped alignment per candidate start, for ~1.5 orphans
// per cluster.  It is planned per cluster, then executed flat:
//   k_plan_rescue         one thread per cluster: the rescue problems TemplateBuilder would pose (result independent).  It, k_select,
//                         k_cluster_sums16 and the two finish kernels of the fragment stage take the clusters by kind
//                         (k_cluster_kinds + k_cluster_order in isaac_gpu.hip), the busiest kinds first
//   k_rescue_windows      one wavefront (and workgroup) per problem slot: the mate's 7-mer table in LDS, the window scanned 64 x 8, 12 or 16
//                         positions at a time, candidate starts collected in a per-problem bitmap (sorted + unique for free)
//   k_rescue_align        one thread per candidate start: UngappedAligner::alignUngapped
//   k_rescue_gapped_plan  one thread per problem: rank of every aligned candidate, the best one, which get a gapped retry
//   k_gapped_jobs         8 lanes per retry, one wavefront of 8 retries per workgroup (bsw_kernel.h)
//   k_predict_heavy       one thread per cluster: which clusters cannot fit the light work lists
//   k_select              one thread per cluster: consumes the rescue results, pair / orphan selection, alignment scores,
//                         clippers, FragmentHeader records
//   k_select_heavy        one wave per predicted cluster, on its own stream next to k_select
static const u32 KMER_EMPTY = 0xffffffffu;
static const u32 RW_TABLE = 512;          // hash slots for the mate's <= 250-odd 7-mers
static const u32 RW_PRESENT_WORDS = 512;  // one bit per possible 7-mer: does the mate have it?
static const u32 RW_MATE_WORDS = 36;      // the mate's bases in LDS, 16 a word (512) and what a read from its last word reads on
static const u32 RW_FLAG_WORDS = 18;      // a bit per mate position, likewise
#ifndef ISAAC_RW_WAVES
#define ISAAC_RW_WAVES 1
#endif
static const u32 RW_WAVES = ISAAC_RW_WAVES;   // wavefronts per workgroup of k_rescue_windows
static const u32 RW_LDS_BITMAP = 64;      // words: windows up to ~1900 bases keep their candidate bitmap in LDS
#ifndef ISAAC_RW_PER_LANE
#define ISAAC_RW_PER_LANE 8
#endif
static const u32 RW_PER_LANE = ISAAC_RW_PER_LANE;        // consecutive window positions per lane and tile (a multiple of 8)
static const i32 RW_TILE = 64 * RW_PER_LANE;              // window positions per wave and tile
static const u32 CAND_REGIONS = 256;

struct RescueBuffers
{
    RescueJob *jobs; u32 jobsCap; u32 *jobCounter;
    u8 *jobActive;                 // per slot: a problem for k_rescue_windows (valid, not given up by the plan).  4 MB a chunk: the kernel's empty wavefronts -- two slots in three -- learn it from L2 instead of the 128-byte record
    u32 *bitmaps; u32 bitmapCap; u32 *bitmapCounter;
    i32 *candPositions; u32 *candJob; Cand *shadowCands; CandSummary *candSummaries; u32 *shadowCigars; u32 *candRank; u32 candCap; u32 *candCounter;
    // candidate slots are handed out from CAND_REGIONS equal regions, each with its own counter (candCounter[region]): one
    // counter for every workgroup of a chunk serialises at ~8 ns per atomic
    u32 candRegionSize;
    u32 *jobBase; u32 *jobCount;   // per cluster of the chunk; jobBase == 0xffffffff: the cluster runs its rescues itself
};

#ifndef ISAAC_SELECT_BLOCK
#define ISAAC_SELECT_BLOCK 64
#endif
static const u32 SELECT_BLOCK = ISAAC_SELECT_BLOCK;     // threads per workgroup of k_select / k_plan_rescue
// candidates of both reads together (and rescue problems) up to which k_select / k_plan_rescue copy a cluster's lists to LDS (0: never); the area of
// one thread in 8-byte words: the candidates, the problems' outcomes (k_select), and a word that makes the count odd
#ifndef ISAAC_SELECT_STAGE
#define ISAAC_SELECT_STAGE 4
#endif
#ifndef ISAAC_PLAN_STAGE
#define ISAAC_PLAN_STAGE 6
#endif
static const u32 SELECT_STAGE = ISAAC_SELECT_STAGE, SELECT_STAGE_WORDS = (SELECT_STAGE * (64 + 24) / 8) | 1u;
static const u32 PLAN_STAGE = ISAAC_PLAN_STAGE, PLAN_STAGE_WORDS = PLAN_STAGE * 8 + 1;

// k_cluster_sums: the outcome of every rescue problem of a cluster and its probability sums (sums.h), one wavefront per cluster
// with room for 64 list entries in LDS; clusters with longer lists are listed for the workgroup-per-cluster form (1024 entries),
// and what neither can do (near ties, lists beyond that, capacity misses of the flat pass) for the wave-per-cluster pass.
static const u32 SUMS_HUGE_ENTRY = 44;   // bytes per list entry of the HBM tier: the key arrays (42) + the second index array of the radix ordering
// SUMS_XL_CAP is not a power of two: its index array is padded to the next one for the sorting network (SUMS_XL_LDS)
static const u32 SUMS_XL_LDS = 3584 * 42 + (4096 - 3584) * 2;
#ifndef ISAAC_SUMS_CLOSE_IDX
#define ISAAC_SUMS_CLOSE_IDX 16384
#endif
static const u32 SUMS_QUARTER_CAP = 16, SUMS_WAVE_CAP = 64, SUMS_MID_CAP = 256, SUMS_BLOCK_CAP = 1024, SUMS_XL_CAP = 3584, SUMS_HUGE_CAP = 65528 /* what 16-bit entry indexes allow; the reference reserves seeds x repeat threshold x 2000 pairs */,
                 SUMS_HUGE_DIGITS = 32768 /* entries whose radix digits fit the LDS array */, SUMS_HUGE_CLOSE_IDX = ISAAC_SUMS_CLOSE_IDX /* entries whose two index arrays fit it too */, SUMS_HUGE_BLOCKS = 512;
struct SumsBuffers { ClusterSums *sums; u8 *residualFlag; u32 *residualList, *residualCount, *mediumList, *mediumCount, *midList, *midCount, *largeList, *largeCount, *xlList, *xlCount, *hugeList, *hugeCount; u8 *hugeKeys; };

static const u32 HEAVY_SORT_LDS = 32768;   // u16 indices: heavyCaps().prob / .pair entries

static const u32 BSW_GROUP_LANES = 8;        // lanes that share one banded Smith-Waterman problem (bsw_kernel.h)
#ifndef ISAAC_BSW_GLOBAL_FLAGS
// 1: the traceback flags of k_gapped_jobs in device memory instead of LDS, five wavefronts per SIMD instead of two.  Measured (2 x 150, 1 M pairs a
// step): 7.08 ms against 4.27 -- the twelve-byte stores of 64 lanes and the traceback's reads through the L2 cost more than the wavefronts bring
// (profiles/exp_r4_bsw_global_flags.log).  Kept as a build switch.
#define ISAAC_BSW_GLOBAL_FLAGS 0
#endif
static const u32 GAPPED_GRID = 32768;        // workgroups of k_gapped_jobs (it strides over the problems)
static const u32 BSW_BLOCK = 64;             // threads per workgroup of k_gapped_jobs / k_bsw_batch: one wavefront, eight problems
// LDS bytes of one banded Smith-Waterman group (bsw_kernel.h)
// traceback flags: 80 bytes per eight rows (bsw_kernel.h)
__host__ __device__ inline u32 bswFlagBytes(u32 maxQueryLength) { return ((maxQueryLength + 7) / 8) * 80; }
__host__ __device__ inline u32 bswGroupLdsBytes(u32 maxQueryLength) { return bswFlagBytes(maxQueryLength) + 128; }
// k_gapped_jobs also keeps the query and the database window of the group there (the DP loop then reads LDS, not global memory)
// (the eight groups of a wave touch their areas at the same offsets in the same instruction: a stride of an odd number of 16-byte units
// spreads them over all LDS banks)
// staged: the sequences too (k_gapped_jobs_staged); else they are in registers and the group's LDS is its end values and traceback flags
__host__ __device__ inline u32 gappedGroupLdsBytes(u32 maxQueryLength, bool staged)
{
    // the end values (128 bytes), [the staged query and database window, each with room for the look-ahead reads,] the traceback flags (ISAAC_BSW_GLOBAL_FLAGS:
    // those in device memory, GAPPED_GRID x groups regions of bswFlagBytes)
    const u32 bytes = 128 + (staged ? 2 * ((maxQueryLength + 47) & ~15u) : 0u) + (ISAAC_BSW_GLOBAL_FLAGS ? 0u : bswFlagBytes(maxQueryLength));
    return (((bytes + 15) / 16) | 1u) * 16;
}
// what the registers of the two forms hold (bsw_kernel.h: gappedJobsBody)
static const u32 BSW_REGISTER_BASES_SHORT = 177, BSW_REGISTER_BASES_LONG = 305;

__global__ __launch_bounds__(64) void k_build_fragments(DevParams P, const u8 *__restrict__ bcl, u32 clusterBase, u32 nChunk, const Match *__restrict__ matches, const u64 *__restrict__ offsets, int trim, ClusterPools pools, AlignList al, u32 *generalList, u32 *generalCount);
__global__ __launch_bounds__(64) void k_build_fragments_general(DevParams P, const u8 *bcl, u32 clusterBase, const Match *matches, const u64 *offsets, int trim, u8 *matchOrderArena, u8 *orderArena, ClusterPools pools, AlignList al, const u32 *list, const u32 *listCount);
__global__ __launch_bounds__(256) void k_adapter_ranges(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, u32 nChunk, ClusterPools pools);
__global__ __launch_bounds__(256) void k_rescue_adapter_ranges(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, RescueBuffers rb);
__global__ __launch_bounds__(256) void k_align_candidates(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, ClusterPools pools, AlignList al, Counters *counters);
__global__ __launch_bounds__(64) void k_finish_candidates(DevParams P, u32 nChunk, int withGaps, u32 *indelList, u32 *indelCount, ClusterPools pools, GappedBuffers gb, const u32 *__restrict__ order, u32 *generalList, u32 *generalCount);
__global__ __launch_bounds__(64) void k_finish_candidates_general(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, int withGaps, u32 *indelList, u32 *indelCount, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *list, const u32 *listCount);
__global__ __launch_bounds__(64) void k_indel_fragments(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, int withGaps, const u32 *indelList, const u32 *indelCount, ClusterPools pools, GappedBuffers gb, Counters *counters);
__global__ __launch_bounds__(64) void k_finish_fragments(DevParams P, u32 nChunk, int withGaps, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *__restrict__ order, u32 *generalList, u32 *generalCount);
__global__ __launch_bounds__(64) void k_finish_fragments_general(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, int withGaps, u32 *tflagsArena, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *list, const u32 *listCount);
__global__ __launch_bounds__(SELECT_BLOCK) void k_plan_rescue(const TemplateConstants *__restrict__ constants, DevReference R, u32 clusterBase, u32 nChunk, ClusterPools pools, RescueBuffers rb, const u32 *__restrict__ order);
__global__ __launch_bounds__(64 * RW_WAVES) void k_rescue_windows(DevParams P, DevReference R, u64 totalBases, const u8 *bcl, u32 clusterBase, RescueBuffers rb);
__global__ __launch_bounds__(256) void k_rescue_align(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, ClusterPools pools, RescueBuffers rb, Counters *counters);
__global__ __launch_bounds__(256) void k_rescue_gapped_plan(ClusterPools pools, RescueBuffers rb, GappedBuffers gb, u32 *longList, u32 *longCount, Counters *counters);
__global__ __launch_bounds__(256) void k_rescue_gapped_plan_long(ClusterPools pools, RescueBuffers rb, GappedBuffers gb, const u32 *longList, const u32 *longCount);
__global__ __launch_bounds__(16 * SUMS16_GROUPS) void k_cluster_sums16(DevParams P, ClusterPools pools, u32 nChunk, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters, const u32 *order);
__global__ __launch_bounds__(256) void k_cluster_sums(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters);
__global__ __launch_bounds__(256) void k_cluster_sums_mid(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters);
__global__ __launch_bounds__(256) void k_cluster_sums_large(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters);
__global__ __launch_bounds__(1024) void k_cluster_sums_xl(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters);
__global__ __launch_bounds__(1024) void k_cluster_sums_huge(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters);
__global__ __launch_bounds__(SELECT_BLOCK) void k_select(const TemplateConstants *__restrict__ constants, DevReference R, double logMismatchQ40, const u8 *__restrict__ bcl, u32 clusterBase, u32 nChunk, u32 tile, ClusterPools pools, RescueBuffers rb, const GappedResult *__restrict__ gappedResults, const ClusterSums *__restrict__ sums, FragmentRecord *__restrict__ records, u32 *__restrict__ cigars, u32 *overflowList, u32 *overflowCount, const u8 *__restrict__ skip, Counters *counters, const u32 *__restrict__ order);
__global__ __launch_bounds__(64) void k_select_heavy(DevParams P, DevReference R, DevTls tls, RogCorrection rog, double logMismatchQ40, const u8 *bcl, u32 clusterBase, u32 nList, const u32 *nListDev, u32 tile, ClusterPools pools, u8 *arena, u64 arenaBytes, TemplateCaps caps, const u32 *list, RescueBuffers rb, const GappedResult *gappedResults, const GappedJob *gappedJobs, const ClusterSums *sums, FragmentRecord *records, u32 *cigars, Counters *counters);
namespace isaac
{
__global__ __launch_bounds__(BSW_BLOCK) void k_bsw_batch(int matchScore, int mismatchScore, int gapOpenScore, int gapExtendScore, const char *sequences, const isaac_bsw_job *jobs, u32 nJobs, u32 maxQueryLength, isaac_bsw_result *results);
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap, u32 maxReadLength, GappedResult *results, u8 *flagsArena);
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs_long(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap, u32 maxReadLength, GappedResult *results, u8 *flagsArena);
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs_staged(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap, u32 maxReadLength, GappedResult *results, u8 *flagsArena);
__global__ __launch_bounds__(256) void k_gapped_rescan(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap, GappedResult *results);
}
