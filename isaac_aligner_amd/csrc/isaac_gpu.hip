// MI355X (gfx950) kernels and the C ABI of include/isaac_gpu.h.
//
// Kernels (DESIGN.md §4 has the mapping, the bound and the algorithmic bytes of each)
//   seed lookup      k_find_matches (8 lanes per cluster, prefix directory + bisection of the resident 32-mer table, ExactMaskMatcher's
//                    rules), k_compact_matches
//   fragment stage   k_build_fragments (candidate positions per cluster) -> k_align_candidates (ungapped alignment per candidate) ->
//                    k_finish_candidates (consolidation per cluster) -> k_indel_fragments (single-indel detector, wave per listed
//                    cluster) -> k_gapped_jobs + k_gapped_rescan (banded Smith-Waterman, 16 lanes per problem; bsw_kernel.h) ->
//                    k_finish_fragments (accept rule, final consolidation)
//   template stage   k_plan_rescue -> k_rescue_windows -> k_rescue_align -> k_rescue_gapped_plan -> k_gapped_jobs ->
//                    k_cluster_sums* -> k_select (thread per cluster) -> k_select_heavy (wave per cluster, twice, counts read on the device); template.h
//   leaves / formats k_bsw_batch, k_tls_samples, k_load_candidates / k_write_candidates, k_fq_* (fastq_kernel.h)
//   index builder    k-mer enumeration + hipCUB radix sort + run analysis (+ 70-permutation neighbour annotation), k_prefix_table
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/functional.hpp>
#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include <deque>
#include "kernels.h"
#include "host_util.h"
#include "fastq_kernel.h"
#include "index_kernels.h"
#include "bam_kernels.h"
#include "bgzf_kernels.h"
#include "deflate_kernels.h"

namespace
{
thread_local std::string g_error;

struct HipError : std::runtime_error { hipError_t code; HipError(hipError_t c, const std::string &w) : std::runtime_error(w), code(c) {} };
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw HipError(e_, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T> struct DevBuf
{
    T *p = nullptr; size_t n = 0;
    DevBuf() = default; DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    void reserve(size_t count)
    {
        if (count <= n) return;
        release();
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        n = count;
        // ISAAC_GPU_TRACE_ALLOC=<MB>: device allocations of at least that size on stderr (element size x count)
        static const long traceFrom = std::getenv("ISAAC_GPU_TRACE_ALLOC") ? std::atol(std::getenv("ISAAC_GPU_TRACE_ALLOC")) : -1;
        if (traceFrom >= 0 && count * sizeof(T) >= size_t(traceFrom) << 20) std::fprintf(stderr, "isaac_gpu alloc %8.1f MB = %zu x %zu\n", double(count * sizeof(T)) / 1048576.0, sizeof(T), count);
    }
    void release() { if (p) { hipFree(p); p = nullptr; n = 0; } }
    ~DevBuf() { release(); }
};

struct KernelTimer { double ms = 0; uint64_t launches = 0; };
} // namespace

namespace isaac_host_resolve { struct Resolver; void destroy(Resolver *r); }      // resolve_host.cpp

struct isaac_gpu_ctx
{
    bool ownsStream = false; u32 cigarExtra = 32;
    u32 orderedClusters = 0; const void *orderedFor = nullptr; u32 orderedBase = 0;   // the chunk c->clusterOrder was last made for by the fragment stage (k_plan_rescue takes it as it is)
    hipStream_t bswStream = nullptr; hipEvent_t bswBegin = nullptr, bswEnd = nullptr;   // ISAAC_GPU_BSW_SIDE_STREAM: the banded SW on a stream of the lowest priority (a measurement)
    int device = 0; hipStream_t stream = nullptr;
    hipStream_t downloadStream = nullptr; std::deque<std::pair<u64, hipEvent_t> > downloads; u64 downloadTicket = 0;      // isaac_gpu_download_async
    isaac_params params; DevParams P;
    // reference
    DevBuf<char> basesOwned; const char *bases = nullptr;
    DevBuf<u64> contigOffset; std::vector<u64> hContigOffset; DevBuf<u8> contigLoaded; std::vector<u8> hContigLoaded; u32 nContigs = 0;
    DevBuf<TableEntry> entries; u64 nKmers = 0; DevBuf<u32> karyotype; bool hasKaryotype = false;
    const TableEntry *entriesBorrowed = nullptr;   // isaac_gpu_set_index_dev: a table owned by the caller (another context, an RCCL receive buffer)
    const TableEntry *tableEntries() const { return entriesBorrowed ? entriesBorrowed : entries.p; }
    DevBuf<DevAdapters> adaptersDev; DevBuf<u32> adapterRanges;      // sequencing adapters (--default-adapters): the list with its 5-mer tables; four ranges per candidate slot's cluster
    DevBuf<u32> flaggedList, resolveChanged; DevBuf<u8> resolveStaging, resolveBack; DevBuf<u64> resolveRanges; DevBuf<Match> resolveMatches; std::vector<char> hostBases; const char *hostBasesGiven = nullptr; std::vector<isaac_host_resolve::Resolver *> resolvers; std::vector<u8> resolverLoaded;      // isaac_gpu_resolve_flagged
    DevBuf<u32> prefixTable; u32 prefixBits = 0; std::vector<u64> maskOffsets;   // entries before each mask of the table (load_index / build_index)
    DevBuf<u32> packedBases, notBase;   // 2-bit copy of the contigs + not-ACGT bitmap (k_rescue_windows reads these)
    const u32 *packedBorrowed = nullptr, *notBaseBorrowed = nullptr, *prefixBorrowed = nullptr;      // isaac_gpu_share_reference on one device: the owner's
    const u32 *packedWords() const { return packedBorrowed ? packedBorrowed : packedBases.p; }
    const u32 *notBaseWords() const { return notBaseBorrowed ? notBaseBorrowed : notBase.p; }
    const u32 *prefixWords() const { return prefixBorrowed ? prefixBorrowed : prefixTable.p; }
    DevBuf<u64> matchBase;
    // run constants of the template kernels in device memory: passed by value they end up as private copies (dynamic indexing)
    DevBuf<TemplateConstants> templateConstants;
    // isaac_gpu_fastq_to_bcl scratch, kept between calls (hipMalloc costs more than the conversion)
    DevBuf<u8> fqIsStart; DevBuf<u64> fqLineStart, fqLineEnd; DevBuf<int> fqSelected; DevBuf<u32> fqLineMap, fqMapBefore, fqIsHeader, fqRecordIndex, fqFirstBad; DevBuf<FqRecord> fqRecords;
    DevBuf<double> logTables;
    // work
    DevBuf<Match> staging; DevBuf<u32> counts, chunkOffsets; DevBuf<u8> cubTemp; DevBuf<u32> contigHits; std::vector<u32> hContigHits;
    // the chunk's candidates: 32 B of ClusterMeta per cluster, one slot per seed match in the candidate pool, the cigar arena (types.h)
    DevBuf<ClusterMeta> clusterMeta; DevBuf<Cand> candPool; DevBuf<u32> cigarArena, cigarNext; ClusterPools pools; DevBuf<u32> fragTflags, poolShort; DevBuf<u8> fragMatchOrder;
    DevBuf<GappedJob> gappedJobs, rescueGappedJobs; DevBuf<GappedResult> gappedResults, rescueGappedResults; DevBuf<u32> gappedBase, gappedCounters;
    DevBuf<u8> heavyArena, clusterKinds; DevBuf<u32> clusterOrder, kindCounts; DevBuf<u32> overflowList; DevBuf<u32> overflowCount; DevBuf<ClusterSums> clusterSums; DevBuf<u32> mediumList, largeList, xlList, hugeList, longJobs; DevBuf<u8> hugeKeys;
    DevBuf<TlsSample> tlsSamples; DevBuf<u32> cigarLengths, cigarOffsets; DevBuf<u64> cigarTotal;
    DevBuf<CrcConstants> crcConstants; bool crcReady = false;    // isaac_gpu_bgzf_store
    DevBuf<u32> binOfContig, binValues, binValuesAlt; DevBuf<u16> binKeys, binKeysAlt; DevBuf<u64> binWords, binCounts, binCuts, binLayout;        // isaac_gpu_bin_tile
    DevBuf<u64> deflateCounts, deflateOffsets; DevBuf<DeflateTables> deflateTables; DevBuf<u8> deflateStaging; DevBuf<u32> deflateSizes;   // isaac_gpu_bgzf_deflate
    // isaac_gpu_bam_records scratch
    DevBuf<BamTile> bamTiles; DevBuf<u64> bamKeyHi, bamKeyLo, bamKeyAlt, bamOffsets, bamBytes64, bamBounds; DevBuf<u32> bamIndex, bamIndexAlt, bamBytes;
    DevBuf<u64> dupPrimary, dupMate, dupRank, dupCluster, dupSmall; DevBuf<u8> dupFlag;        // duplicate marking
    DevBuf<FragmentRecord> realignRecords; DevBuf<RealignGap> realignGaps, realignDeletionEnds; DevBuf<u32> realignPool, realignNext, realignList; DevBuf<u8> realignChanged;   // gap realignment
    DevBuf<RescueJob> jobs; DevBuf<u8> jobActive; DevBuf<u32> rescueCounters, bitmaps, candJob, shadowCigars, jobBase, jobCount; DevBuf<i32> candPositions; DevBuf<Cand> shadowCands; DevBuf<CandSummary> candSummaries; DevBuf<u32> candRank;
    DevBuf<Counters> counters, countersSaved; DevBuf<u8> bswFlags;
    std::map<std::string, KernelTimer> timers;
    struct PendingTimer { std::string name; hipEvent_t e0, e1; };
    std::vector<PendingTimer> pendingTimers; std::vector<hipEvent_t> eventPool;
    // one chunk of a select call as the wave-per-cluster pass (k_select_heavy) sees it
    struct ChunkDesc { const uint8_t *bcl = nullptr; u32 clusterBase = 0, tile = 0; FragmentRecord *records = nullptr; u32 *cigars = nullptr; DevTls tls; RogCorrection rog; };
    bool deferredCompletion = false;
    u32 selectCapacity = 0;        // chunk size the buffers of the select stage were last sized for
    DevBuf<u32> midList, heavyList, heavyCount, indelList, alignList, generalList, generalCount; DevBuf<u8> heavyFlag;
    u32 chunkClusters = 1048576;   // upper bound of a chunk (ISAAC_GPU_CHUNK_CLUSTERS)
    u32 chunkNow = 0;              // the chunk size in use: the largest call so far, rounded up, at most chunkClusters; sizes the chunk-private buffers

    DevReference ref() const
    {
        DevReference r; std::memset(&r, 0, sizeof(r));
        r.bases = bases; r.totalBases = hContigOffset.empty() ? 0 : hContigOffset[nContigs]; r.contigOffset = contigOffset.p; r.contigLoaded = contigLoaded.p; r.nContigs = nContigs;
        r.entries = tableEntries(); r.nKmers = nKmers; r.karyotype = hasKaryotype ? karyotype.p : nullptr;
        r.logMatch = logTables.p; r.logMismatch = logTables.p + 100; r.logStride = 1;
        r.prefixTable = prefixBits ? prefixWords() : nullptr; r.prefixBits = prefixBits;
        r.packedBases = packedWords(); r.notBase = notBaseWords();
        return r;
    }
};

namespace
{
// times one kernel launch sequence with HIP events on the stream it is launched on.  The events are only read when the
// timers are queried (resolveTimers), so timing does not put a host synchronisation between the launches.
static hipEvent_t takeEvent(isaac_gpu_ctx *c)
{
    if (!c->eventPool.empty()) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
    hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}
static void resolveTimers(isaac_gpu_ctx *c);
struct ScopedTimer
{
    isaac_gpu_ctx *c; const char *name; hipStream_t st; hipEvent_t e0, e1;
    ScopedTimer(isaac_gpu_ctx *ctx, const char *n, hipStream_t stream = nullptr) : c(ctx), name(n), st(stream ? stream : ctx->stream)
    { e0 = takeEvent(c); e1 = takeEvent(c); hipEventRecord(e0, st); }
    ~ScopedTimer() { hipEventRecord(e1, st); c->pendingTimers.push_back({name, e0, e1}); if (c->pendingTimers.size() > 4096) resolveTimers(c); }
};
static void resolveTimers(isaac_gpu_ctx *c)
{
    for (auto &p : c->pendingTimers)
    {
        hipEventSynchronize(p.e1);
        float ms = 0; hipEventElapsedTime(&ms, p.e0, p.e1);
        KernelTimer &t = c->timers[p.name]; t.ms += ms; ++t.launches;
        c->eventPool.push_back(p.e0); c->eventPool.push_back(p.e1);
    }
    c->pendingTimers.clear();
}


// ------------------------------------------------------------------------------------------------------------------
// k_find_matches: ClusterSeedGenerator::generateThread (ClusterSeedGenerator.cpp:138-192) + ExactMaskMatcher::matchMask
// (ExactMaskMatcher.cpp:83-184) for both seed iterations of FindMatchesTransition::findLaneMatches (:391-427).
// 8 lanes cooperate on one cluster; each lane owns one (seed, strand) probe per round.
#ifndef ISAAC_FIND_POSITION_ON_HIT
#define ISAAC_FIND_POSITION_ON_HIT 1
#endif
#ifndef ISAAC_FIND_GROUP
#define ISAAC_FIND_GROUP 8
#endif
static const u32 FIND_GROUP = ISAAC_FIND_GROUP, FIND_BLOCK = 256, FIND_CLUSTERS_PER_BLOCK = FIND_BLOCK / FIND_GROUP;      // (lanes per cluster: a power of two)
#ifndef ISAAC_FIND_SLICE
#define ISAAC_FIND_SLICE 4
#endif
static const u32 FIND_SLICE = ISAAC_FIND_SLICE;       // slices of the table of up to this many entries are fetched whole (0: always the bisection)

__device__ inline u64 lowerBound(const TableEntry *entries, u64 lo, u64 hi, u64 key, u32 &steps)
{
    while (lo < hi)
    {
        const u64 mid = (lo + hi) >> 1;
        if (entries[mid].kmer < key) lo = mid + 1; else hi = mid;
        ++steps;
    }
    return lo;
}

// Round 6, three attempts on the second seed iteration (16 % of the probes -- the reads the first iteration left open, mostly reads of repeat families whose
// probes bisect buckets of hundreds of entries -- and 40 % of the kernel: one iteration alone takes 0.69 ms, both 1.15), all measured slower or equal and taken
// out again: (1) the workgroup's second-iteration probes listed in LDS and dealt out densely, a thread per probe: 1.11 ms -- the waves that have nothing to do
// still wait at the workgroup's barriers; (2) an eight-way search (seven pivots a round trip) instead of the bisection: 1.40 ms -- the kernel runs at 62 % of
// the chip's rate of random lines (49.5 G lines/s whatever the access shape, profiles/exp_r6_random_lines.log) and seven lines where the bisection touches
// one cost more than the shorter chain brings; (3) the second iteration as a kernel of its own over a list of the clusters with an open read: 1.77 ms -- as a
// second round of this kernel its long chains run beside other workgroups' first rounds, alone they run by themselves.  profiles/exp_r6_find.log
// starts[b] = first table index whose k-mer has leading bits >= b (b = 0 .. 2^bits, the last one = n)
__global__ void k_prefix_table(const TableEntry *kmers, u64 n, u32 bits, u32 *starts)
{
    const u64 b = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b > (u64(1) << bits)) return;
    u32 steps = 0;
    starts[b] = (b >> bits) ? u32(n) : u32(lowerBound(kmers, 0, n, b << (64 - bits), steps));
}
// The directory k_find_matches reads: two words a bucket, the bucket's first table index and a fingerprint of each of its first four entries -- the eight bits of the
// k-mer behind the bucket's own.  A probe whose fingerprint is not among those of a bucket of up to four entries has no match (the bits differ, so does the k-mer)
// and ends at the directory's line without touching the table: half the probes and more, every read is looked up on both strands.
#ifndef ISAAC_FIND_FINGERPRINTS
#define ISAAC_FIND_FINGERPRINTS 1
#endif
__device__ inline u32 kmerFingerprint(u64 kmer, u32 bits) { return u32(kmer >> (56 - bits)) & 0xffu; }
__global__ void k_prefix_directory(const TableEntry *kmers, u32 bits, const u32 *starts, u32 *directory)
{
    const u64 b = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b > (u64(1) << bits) + 1) return;
    u32 first = 0xffffffffu, prints = 0;
    if (b <= (u64(1) << bits))
    {
        first = starts[b];
        const u32 end = (b >> bits) ? first : starts[b + 1];
        for (u32 i = 0; i < 4 && first + i < end; ++i) prints |= kmerFingerprint(kmers[first + i].kmer, bits) << (8 * i);
    }
    directory[2 * b] = first; directory[2 * b + 1] = prints;
}

#ifndef ISAAC_WAVES_FIND
#define ISAAC_WAVES_FIND 0      // (measurements: scripts/exp_r6_find_waves.sh builds the kernel for eight and for four waves per SIMD)
#endif
#if ISAAC_WAVES_FIND
__attribute__((amdgpu_waves_per_eu(ISAAC_WAVES_FIND, ISAAC_WAVES_FIND)))
#endif
__global__ __launch_bounds__(FIND_BLOCK) void k_find_matches(DevParams P, DevReference R, const u8 *bcl, u32 nClusters, u32 clusterBase, u32 tile,
                                                             Match *staging, u32 *counts, u32 stride, u32 *contigHits, Counters *counters)
{
    extern __shared__ __align__(16) u8 sbcl[];
    const u32 CL = P.clusterLength;
    const u32 firstCluster = blockIdx.x * FIND_CLUSTERS_PER_BLOCK;
    STAMP_BEGIN();
    {   // stage the block's clusters (contiguous bytes) through LDS with coalesced loads
        const u64 base = u64(firstCluster) * CL;
        const u32 nBytes = imin<u32>(FIND_CLUSTERS_PER_BLOCK, nClusters - firstCluster) * CL;
        const u8 *src = bcl + base;
        if (((reinterpret_cast<uintptr_t>(src)) & 15) == 0)
        {
            const u32 nVec = nBytes / 16;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src); uint4 *d4 = reinterpret_cast<uint4 *>(sbcl);
            for (u32 i = threadIdx.x; i < nVec; i += FIND_BLOCK) d4[i] = s4[i];
            for (u32 i = nVec * 16 + threadIdx.x; i < nBytes; i += FIND_BLOCK) sbcl[i] = src[i];
        }
        else for (u32 i = threadIdx.x; i < nBytes; i += FIND_BLOCK) sbcl[i] = src[i];
    }
    __syncthreads();
    STAMP(10);
    const u32 group = threadIdx.x / FIND_GROUP, lane = threadIdx.x % FIND_GROUP;
    const u32 cluster = firstCluster + group;
    const bool active = cluster < nClusters;
    const u8 *cb = sbcl + group * CL;
    Counters local; memset(&local, 0, sizeof(local));
    u32 complete = 0;      // bit r: read r complete (TileClusterInfo.hh:65-209)
    u32 written = 0;       // records of this cluster so far (uniform over the group)
    Match *out = staging + u64(cluster) * stride;
    for (u32 pass = 0; pass < 2; ++pass)
    {
        const u32 nProbes = 2 * P.nPass[pass];
        u32 completeNext = complete;
        for (u32 base = 0; base < nProbes; base += FIND_GROUP)
        {
            const u32 p = base + lane;
            bool probe = active && p < nProbes;
            u32 seedIdx = 0, strand = 0, readIdx = 0;
            if (probe)
            {
                seedIdx = P.passSeeds[pass][p >> 1]; strand = p & 1; readIdx = P.seeds[seedIdx].readIndex;
                if ((complete >> readIdx) & 1) probe = false;      // ClusterSeedGenerator.cpp:148
            }
            u32 nrec = 0; u64 first = 0, pos0 = 0; bool tooMany = false; bool completes = false;
            if (probe)
            {
                // k-mer of the seed: forward MSB-first, reverse complement built from the other end (ClusterSeedGenerator.cpp:162-176)
                const u8 *b = cb + P.readOffset[readIdx] + P.seeds[seedIdx].offset;
                u64 kmer = 0; bool isN = false;
                #pragma unroll 8
                for (u32 i = 0; i < 32; ++i)
                {
                    const u32 v = b[i];
                    isN |= !(v & 0xfc);
                    if (strand) kmer = (kmer >> 2) | (u64((~v) & 3) << 62); else kmer = (kmer << 2) | (v & 3);
                }
                STAMP(11);
                if (!isN)
                {
                    ++local.probes;
                    u32 steps = 0;
                    u32 r = 0;
                    u64 lo = 0, hi = R.nKmers;
                    if (R.prefixTable)
                    {   // the k-mer's leading bits select a slice of the table: one 8-byte read instead of most of the bisection
                        const u64 bucket = kmer >> (64 - R.prefixBits);
                        const uint4 range = *reinterpret_cast<const uint4 *>(R.prefixTable + 2 * bucket);   // [bucket], [bucket + 1]: first index, fingerprints
                        ++steps;
                        lo = range.x; hi = range.z;
#if ISAAC_FIND_FINGERPRINTS
                        if (hi - lo <= 4)
                        {   // is the probe's fingerprint among the bucket's?  (x ^ pattern has a zero byte; bytes past the bucket's end do not count)
                            const u32 x = range.y ^ (kmerFingerprint(kmer, R.prefixBits) * 0x01010101u);
                            const u32 zeroBytes = (x - 0x01010101u) & ~x & 0x80808080u;
                            const u32 inBucket = hi == lo ? 0u : 0x80808080u >> (8 * (4 - u32(hi - lo)));
                            if (!(zeroBytes & inBucket)) hi = lo;
                        }
#endif
                    }
                    // ExactMaskMatcher.cpp:118-126: reference entries with the same k-mer, at most repeatThreshold of them.
#if ISAAC_FIND_SLICE
                    if (hi - lo <= FIND_SLICE)
                    {   // The usual slice (the directory has about as many buckets as the table has entries): all of it asked for at once -- k-mers and
                        // positions, 16 bytes an entry as the mask files hold them -- and searched in registers: one round trip where the bisection, the
                        // look at the entries behind it and the position were three or four, one after the other
                        TableEntry e[FIND_SLICE];
#pragma unroll
                        for (u32 i = 0; i < FIND_SLICE; ++i)
                        {
                            const uint4 v = lo + i < hi ? *reinterpret_cast<const uint4 *>(R.entries + lo + i) : make_uint4(0xffffffffu, 0xffffffffu, 0, 0);
                            e[i].kmer = u64(v.x) | (u64(v.y) << 32); e[i].position = u64(v.z) | (u64(v.w) << 32);
                        }
                        if (hi > lo) ++steps;
                        u32 before = 0;                              // entries of the slice below the k-mer (the slice is sorted)
#pragma unroll
                        for (u32 i = 0; i < FIND_SLICE; ++i) { const bool in = lo + i < hi; before += (in && e[i].kmer < kmer) ? 1u : 0u; r += (in && e[i].kmer == kmer) ? 1u : 0u; }
                        first = lo + before;
#pragma unroll
                        for (u32 i = 0; i < FIND_SLICE; ++i) if (i == before) pos0 = e[i].position;          // (no register array is indexed at run time)
                        if (!r) pos0 = 0;
                    }
                    else
#endif
                    {
                        first = lowerBound(R.entries, lo, hi, kmer, steps);
                        // the first two entries behind the bisection together: most hits are single entries
                        const bool in0 = first < R.nKmers, in1 = first + 1 < R.nKmers;
                        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
                        if (in0) v0 = *reinterpret_cast<const uint4 *>(R.entries + first);
                        if (in1) v1 = *reinterpret_cast<const uint4 *>(R.entries + first + 1);
                        const u64 k0 = u64(v0.x) | (u64(v0.y) << 32), k1 = u64(v1.x) | (u64(v1.y) << 32);
                        pos0 = (in0 && k0 == kmer) ? (u64(v0.z) | (u64(v0.w) << 32)) : 0;
                        if (in0 && k0 == kmer)
                        {
                            r = 1;
                            if (in1 && k1 == kmer) { r = 2; while (first + r < R.nKmers && r < P.repeatThreshold && R.entries[first + r].kmer == kmer) ++r; }
                        }
                    }
                    local.probeSteps += steps;
                    STAMP(12);
                    if (r)
                    {
                        if (r >= P.repeatThreshold || refposIsTooMany(pos0))
                        {   // :135-154 generateTooManyMatches; the second iteration closes the read (MatchFinder.cpp:299)
                            tooMany = true; nrec = 1;
                            if (pass == 1) completes = true;
                        }
                        else
                        {   // :157-170
                            nrec = r;
                            completes = P.ignoreNeighbors || !(pos0 & 1);
                        }
                    }
                }
            }
            STAMP(13);
            // exclusive prefix of nrec over the 8 lanes of the group
            u32 incl = nrec;
            for (u32 o = 1; o < FIND_GROUP; o <<= 1) { const u32 t = __shfl_up(incl, o, FIND_GROUP); if (lane >= o) incl += t; }
            const u32 total = __shfl(incl, FIND_GROUP - 1, FIND_GROUP);
            u32 at = written + incl - nrec;
            if (nrec)
            {
                const u64 sid = seedId(tile, 0, clusterBase + cluster, seedIdx, strand);
                if (tooMany) { if (at < stride) { out[at].seedId = sid; out[at].location = 0; } }
                else for (u32 i = 0; i < nrec; ++i, ++at)
                {
                    u64 pos = i ? R.entries[first + i].position : pos0;
                    if (R.karyotype) { const u32 c = u32(pos >> 41); pos = (u64(R.karyotype[c - 1] + 1) << 41) | (pos & ((u64(1) << 41) - 1)); }
                    if (at < stride) { out[at].seedId = sid; out[at].location = pos; }
                    // MatchDistribution::addMatches (:173-181): the contig is not empty.  Read before write: millions of stores to
                    // the same word serialise in L2, reads of it do not
                    if (!contigHits[refposContig(pos)]) contigHits[refposContig(pos)] = 1;
                }
                local.matches += nrec;
            }
            STAMP(14);
            written += total;
            u32 c = completes ? (1u << readIdx) : 0;
            for (u32 o = 1; o < FIND_GROUP; o <<= 1) c |= __shfl_xor(c, o, FIND_GROUP);
            completeNext |= c;
        }
        complete = completeNext;
    }
    if (active && lane == 0) counts[cluster] = imin(written, stride);
    STAMP(15);
    flushCounters(local, counters);
    STAMP(16);
}

// the running output offset stays on the device: no host round trip between the chunks of a tile
__global__ void k_advance_match_base(u64 *baseDev, const u32 *chunkOffsets, const u32 *counts, u32 nClusters, u64 *offsetsEnd)
{
    *baseDev += u64(chunkOffsets[nClusters - 1]) + counts[nClusters - 1];
    *offsetsEnd = *baseDev;
}

__global__ void k_compact_matches(const Match *staging, const u32 *counts, const u32 *chunkOffsets, u32 nClusters, u32 stride, const u64 *baseDev,
                                  Match *out, u64 capacity, u64 *offsetsOut)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nClusters) return;
    const u64 at = *baseDev + chunkOffsets[c];
    offsetsOut[c] = at;
    const u32 n = counts[c];
    const Match *src = staging + u64(c) * stride;
    for (u32 i = 0; i < n; ++i) if (at + i < capacity) out[at + i] = src[i];
}

__global__ void k_tls_samples(ClusterPools pools, const u64 *offsets, u32 clusterBase, u32 nChunk, TlsSample *samples)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nChunk) return;
    clusterTlsSample(clusterView(pools.meta[t], pools.cands, pools.cigars), u32(offsets[clusterBase + t + 1] - offsets[clusterBase + t]), samples[t]);
}

// candidates of the chunk -> compact ABI records; one thread per cluster, offsets from an exclusive scan of the counts
// k_plan_rescue and k_select give a cluster to a thread, and what a thread does depends on its cluster's candidate lists: a wave of
// clusters as they come has a quarter of its lanes at work on average, and every lane's private memory access touches a cache line of
// its own.  The clusters are therefore handed out by kind -- candidates per read and, for k_select, rescue problems -- so that the
// lanes of a wave mostly take the same branches and make the same number of turns.
#ifndef ISAAC_PLAN_ORDER
#define ISAAC_PLAN_ORDER 1      // 0: k_plan_rescue takes the clusters as they come (A/B builds)
#endif
#ifndef ISAAC_CLUSTER_ORDER
#define ISAAC_CLUSTER_ORDER 1
#endif
static const u32 CLUSTER_KINDS = 256, KIND_BLOCK = 256;         // small workgroups: beside the kernels of other contexts a 1 024-thread workgroup waits long for a CU with 16 free wave slots
// A counting sort in two launches (a radix sort of the library is twenty launches and a third of a millisecond for these 1 M keys, four
// times per step): the kinds and their histogram, then every cluster's place -- the kind's first place, the block's share of the kind
// (one atomic per block and kind in use), the cluster's rank among the block's clusters of the kind (LDS).  The order inside a kind
// is the order the blocks arrive in, which nothing depends on.
__global__ __launch_bounds__(KIND_BLOCK) void k_cluster_kinds(ClusterPools pools, u32 nChunk, const u32 *jobCount, u8 *kinds, u32 *histogram)
{
    __shared__ u32 h[CLUSTER_KINDS];
    if (threadIdx.x < CLUSTER_KINDS) h[threadIdx.x] = 0;
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nChunk)
    {
        const ClusterMeta m = pools.meta[t];
        u32 kind = 0;
        if (m.built) kind = 1 + (imin<u32>(m.nCands[0], 5) * 6 + imin<u32>(m.nCands[1], 5)) * 4 + (jobCount ? imin<u32>(jobCount[t], 3) : 0);
        // The kinds with most to do first, so that the grid's last waves are short ones (the other way round the order gains half as much, and
        // k_rescue_windows, whose problem slots follow k_plan_rescue's order, loses).  A finer key -- candidates of both reads together up to 31,
        // problems up to 7, read 0's candidates up to 15 -- was worse (k_select 3.0 against 2.7 ms): the split between the reads matters most.
        kind = CLUSTER_KINDS - 1 - kind;
        kinds[t] = u8(kind);
        atomicAdd(&h[kind], 1u);
    }
    __syncthreads();
    if (threadIdx.x < CLUSTER_KINDS && h[threadIdx.x]) atomicAdd(&histogram[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(KIND_BLOCK) void k_cluster_order(const u8 *kinds, u32 nChunk, const u32 *histogram, u32 *cursor, u32 *order)
{
    __shared__ u32 first[CLUSTER_KINDS], h[CLUSTER_KINDS], share[CLUSTER_KINDS];
    if (threadIdx.x < CLUSTER_KINDS) { first[threadIdx.x] = histogram[threadIdx.x]; h[threadIdx.x] = 0; }
    __syncthreads();
    for (u32 o = 1; o < CLUSTER_KINDS; o <<= 1)                              // inclusive sums of the histogram
    {
        u32 v = 0;
        if (threadIdx.x < CLUSTER_KINDS && threadIdx.x >= o) v = first[threadIdx.x - o];
        __syncthreads();
        if (threadIdx.x < CLUSTER_KINDS) first[threadIdx.x] += v;
        __syncthreads();
    }
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 kind = 0, rank = 0;
    if (t < nChunk) { kind = kinds[t]; rank = atomicAdd(&h[kind], 1u); }
    __syncthreads();
    if (threadIdx.x < CLUSTER_KINDS) share[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u;
    __syncthreads();
    if (t < nChunk) order[(kind ? first[kind - 1] : 0u) + share[kind] + rank] = t;
}

__global__ void k_count_candidates(ClusterPools pools, u32 nChunk, u32 *nCands, u32 *nCigar)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nChunk) return;
    const ClusterFragments f = clusterView(pools.meta[t], pools.cands, pools.cigars);
    u32 n = f.nCands[0] + f.nCands[1], c = 0;
    for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) c += f.cands[r][i].cigarLength;
    nCands[t] = n; nCigar[t] = c;
}
__global__ void k_write_candidates(ClusterPools pools, u32 clusterBase, u32 nChunk, const u32 *candOffsets, const u32 *cigarOffsets, u64 candBase, u64 cigarBase,
                                   isaac_candidate *out, u64 capacity, u32 *cigarOut, u64 cigarCapacity)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nChunk) return;
    const ClusterFragments f = clusterView(pools.meta[t], pools.cands, pools.cigars);
    u64 at = candBase + candOffsets[t], cat = cigarBase + cigarOffsets[t];
    for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i, ++at)
    {
        const Cand &k = f.cands[r][i];
        if (at < capacity && cat + k.cigarLength <= cigarCapacity)
        {
            isaac_candidate o; memset(&o, 0, sizeof(o));
            o.position = k.position; o.log_probability = k.logProbability; o.cluster = clusterBase + t; o.read_index = k.readIndex; o.contig_id = k.contigId;
            o.observed_length = k.observedLength; o.reverse = k.reverse; o.mismatch_count = k.mismatchCount; o.matches_in_a_row = k.matchesInARow; o.gap_count = k.gapCount;
            o.edit_distance = k.editDistance; o.smith_waterman_score = k.smithWatermanScore; o.unique_seed_count = k.uniqueSeedCount;
            o.non_unique_first = k.nonUniqueFirst == NON_UNIQUE_NONE ? 0xffffffffu : k.nonUniqueFirst; o.non_unique_second = k.nonUniqueSecond;
            o.repeat_seeds_count = k.repeatSeedsCount; o.cigar_offset = u32(cat); o.cigar_length = k.cigarLength; o.low_clipped = k.lowClipped; o.high_clipped = k.highClipped;
            o.first_seed_index = k.firstSeedIndex;
            out[at] = o;
            for (u32 w = 0; w < k.cigarLength; ++w) cigarOut[cat + w] = f.cigarPool[k.cigarOffset + w];
        }
        cat += k.cigarLength;
    }
}

// the inverse of k_write_candidates: the chunk's pools from caller-supplied candidate lists (isaac_gpu_select_candidates): a cluster owns
// as many slots as the caller lists candidates for it, read 0's before read 1's
__global__ void k_load_candidates(DevParams P, const u8 *bcl, u32 clusterBase, u32 nChunk, const isaac_candidate *cands, const u64 *candOffsets, const u32 *cigarIn, int trim,
                                  ClusterPools pools)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nChunk) return;
    const u64 chunkBegin = candOffsets[clusterBase], begin = candOffsets[clusterBase + t], end = candOffsets[clusterBase + t + 1];
    const u32 first = u32(begin - chunkBegin);
    const u32 cap = (u64(first) + (end - begin) <= pools.candCap && end - begin <= 2 * CAND_CAP) ? u32(end - begin) : 0u;
    ClusterFragments f = clusterViewNew(first, cap, pools.cands, pools.cigars);
    if (cap != end - begin) f.flags |= CLUSTER_OVERFLOW;
    const u8 *clusterBcl = bcl + u64(clusterBase + t) * P.clusterLength;
    for (u32 r = 0; r < 2; ++r)
        f.endCyclesMasked[r] = (trim && r < P.nReads) ? trimLowQualityEnd(clusterBcl + P.readOffset[r], P.readLength[r], P.baseQualityCutoff) : 0;
    u32 words = 0;
    for (u64 i = begin; i < begin + cap; ++i) words += cands[i].cigar_length;
    if (words > f.cigarCap) clusterCigarExtra(f, pools.cigars, 3 * pools.candCap + atomicAdd(pools.cigarNext, words), words, pools.cigarCap);
    CigarPool pool; pool.words = f.cigarPool; pool.used = f.cigarUsed; pool.capacity = f.cigarCap; pool.overflow = 0;
    for (u32 r = 0; r < 2; ++r)
    {
        if (r) { f.cands[1] = f.cands[0] + f.nCands[0]; f.candCap[1] = f.candCap[0] - f.nCands[0]; }
        for (u64 i = begin; i < begin + cap; ++i)
        {
            const isaac_candidate &o = cands[i];
            if ((o.read_index & 1) != r) continue;
            if (f.nCands[r] == CAND_CAP) { f.flags |= CLUSTER_OVERFLOW; continue; }
            Cand &k = f.cands[r][f.nCands[r]++];
            candInit(k, r);
            k.position = o.position; k.logProbability = o.log_probability; k.contigId = o.contig_id; k.observedLength = o.observed_length; k.reverse = o.reverse != 0;
            k.mismatchCount = u16(o.mismatch_count); k.matchesInARow = u16(o.matches_in_a_row); k.gapCount = u16(o.gap_count); k.editDistance = u16(o.edit_distance);
            k.smithWatermanScore = o.smith_waterman_score; k.uniqueSeedCount = u16(o.unique_seed_count);
            k.nonUniqueFirst = o.non_unique_first == 0xffffffffu ? NON_UNIQUE_NONE : u16(o.non_unique_first); k.nonUniqueSecond = u16(o.non_unique_second);
            k.repeatSeedsCount = u16(o.repeat_seeds_count); k.lowClipped = u16(o.low_clipped); k.highClipped = u16(o.high_clipped); k.firstSeedIndex = (signed char)o.first_seed_index;
            k.cigarOffset = pool.used; k.cigarLength = u16(o.cigar_length);
            for (u32 w = 0; w < o.cigar_length; ++w) pool.push(cigarIn[o.cigar_offset + w]);
            f.repeatSeedsCount = o.repeat_seeds_count;
            f.built = 1;
        }
    }
    f.cigarUsed = pool.used;
    if (pool.overflow) f.flags |= CLUSTER_OVERFLOW;
    clusterViewStore(f, pools.cands, pools.meta[t]);
}

u32 gridFor(u64 n, u32 block)
{
    // a launch holds fewer than 2^32 work-items (the dispatch packet counts them in 32 bits): anything that can be larger must use gridStrided
    const u64 blocks = (n + block - 1) / block;
    if (blocks * block >= (u64(1) << 32)) throw std::invalid_argument("launch of 2^32 work-items or more");
    return u32(blocks);
}
// for grid-stride kernels over arrays that may hold 2^32 elements and more
u32 gridStrided(u64 n, u32 block) { return u32(std::min<u64>((n + block - 1) / block, (u64(1) << 22))); }

// scans and sorts: rocPRIM called directly (rounds 1-5 went through hipCUB, the CUB-shaped wrapper around it, whose entry points take `int` counts)
inline void checkCount(size_t n, const char *what)
{   // the wrappers below serve per-chunk and per-mask arrays whose elements are addressed with 32-bit indexes by the kernels around them (a chunk has at most 2^20
    // clusters, a mask of the GRCh38 table 46 M entries): a count that does not fit is refused here, loudly, instead of wrapping there.  (The one sort that is
    // larger -- the distinct k-mers of the neighbour annotation -- calls rocPRIM with its size_t count itself.)
    if (n >= (size_t(1) << 31)) throw std::length_error(std::string(what) + ": 2^31 or more items in one device scan / sort");
}
template <typename T> void exclusiveSum(isaac_gpu_ctx *c, const T *in, T *out, size_t n)
{
    checkCount(n, "exclusiveSum");
    size_t bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
    c->cubTemp.reserve(bytes + 16);
    HIP_CHECK(rocprim::exclusive_scan(c->cubTemp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
}
template <typename T> void inclusiveSum(isaac_gpu_ctx *c, const T *in, T *out, size_t n)
{
    checkCount(n, "inclusiveSum");
    size_t bytes = 0;
    HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, rocprim::plus<T>(), c->stream));
    c->cubTemp.reserve(bytes + 16);
    HIP_CHECK(rocprim::inclusive_scan(c->cubTemp.p, bytes, in, out, n, rocprim::plus<T>(), c->stream));
}
template <typename K, typename V> void sortPairs(isaac_gpu_ctx *c, const K *kin, K *kout, const V *vin, V *vout, size_t n, int endBit = int(sizeof(K) * 8))
{
    checkCount(n, "sortPairs");
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0u, unsigned(endBit), c->stream));
    c->cubTemp.reserve(bytes + 16);
    HIP_CHECK(rocprim::radix_sort_pairs(c->cubTemp.p, bytes, kin, kout, vin, vout, n, 0u, unsigned(endBit), c->stream));
}

int fail(int code, const std::string &what) { g_error = what; return code; }

// the clusters of a chunk by kind (k_cluster_kinds): c->clusterOrder
static const u32 *orderClustersByKind(isaac_gpu_ctx *c, u32 n, const u32 *jobCount)
{
    c->clusterKinds.reserve(c->chunkNow); c->clusterOrder.reserve(c->chunkNow); c->kindCounts.reserve(2 * CLUSTER_KINDS);
    HIP_CHECK(hipMemsetAsync(c->kindCounts.p, 0, 2 * CLUSTER_KINDS * sizeof(u32), c->stream));
    k_cluster_kinds<<<gridFor(n, KIND_BLOCK), KIND_BLOCK, 0, c->stream>>>(c->pools, n, jobCount, c->clusterKinds.p, c->kindCounts.p);
    k_cluster_order<<<gridFor(n, KIND_BLOCK), KIND_BLOCK, 0, c->stream>>>(c->clusterKinds.p, n, c->kindCounts.p, c->kindCounts.p + CLUSTER_KINDS, c->clusterOrder.p);
    HIP_CHECK(hipGetLastError());
    return c->clusterOrder.p;
}

// Chunk size for a call over nClusters clusters.  Kernel durations end in a tail set by their slowest waves, so fewer, larger
// launches are faster; the buffers grow with the largest call seen instead of being sized for the upper bound at once.
u32 chunkFor(isaac_gpu_ctx *c, u32 nClusters)
{
    const u32 wanted = std::min<u32>(c->chunkClusters, std::max<u32>(1024u, u32((u64(nClusters) + 65535) / 65536 * 65536)));
    c->chunkNow = std::max(c->chunkNow, std::min(wanted, c->chunkClusters));
    return c->chunkNow;
}

// after the table changed: the prefix directory of k_find_matches (tables of 2^32 entries and more go without)
void buildPrefixTable(isaac_gpu_ctx *c)
{
    c->prefixBits = 0; c->prefixBorrowed = nullptr;
    if (!c->nKmers || c->nKmers >= (u64(1) << 32) || getenv("ISAAC_GPU_NO_PREFIX_TABLE")) return;
    u32 bits = 16; while (bits < 30 && (u64(1) << bits) < c->nKmers) ++bits;     // about one entry per bucket (human: 2.7), 512 KB .. 8 GB
    const u64 entries = (u64(1) << bits) + 1;
    DevBuf<u32> starts; starts.reserve(entries);
    c->prefixTable.reserve(2 * (entries + 1));                                      // + 1: the last bucket reads a pair
    k_prefix_table<<<gridFor(entries, 256), 256, 0, c->stream>>>(c->tableEntries(), c->nKmers, bits, starts.p);
    HIP_CHECK(hipGetLastError());
    k_prefix_directory<<<gridFor(entries + 1, 256), 256, 0, c->stream>>>(c->tableEntries(), bits, starts.p, c->prefixTable.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->prefixBits = bits;
}
#define ISAAC_TRY try {
#define ISAAC_CATCH } catch (const HipError &e) { return fail(e.code == hipErrorOutOfMemory ? ISAAC_GPU_ENOMEM : ISAAC_GPU_EHIP, e.what()); } \
                      catch (const std::invalid_argument &e) { return fail(ISAAC_GPU_EINVAL, e.what()); } \
                      catch (const std::exception &e) { return fail(ISAAC_GPU_EHIP, e.what()); }
} // namespace

extern "C" {

const char *isaac_gpu_last_error(void) { return g_error.c_str(); }

// the adapter list of the parameters on the device; c->P.adapters stays NULL for an empty list (what every kernel's adapter step tests)
static void setAdapters(isaac_gpu_ctx *c)
{
    c->P.adapters = nullptr; c->P.adapterRanges = nullptr; c->P.adapterCandBase = nullptr;
    if (!c->params.n_adapters) return;
    const DevAdapters host = makeDevAdapters(c->params);
    c->adaptersDev.reserve(1);
    HIP_CHECK(hipMemcpy(c->adaptersDev.p, &host, sizeof(host), hipMemcpyHostToDevice));
    c->P.adapters = c->adaptersDev.p;
    if (c->candPool.p) { c->adapterRanges.reserve(4 * c->candPool.n); c->P.adapterRanges = c->adapterRanges.p; c->P.adapterCandBase = c->candPool.p; }
}

int isaac_gpu_create(int device, const isaac_params *params, void *stream, isaac_gpu_ctx **out)
{
    ISAAC_TRY
    if (!params || !out) return fail(ISAAC_GPU_EINVAL, "null argument");
    int nDevices = 0;
    HIP_CHECK(hipGetDeviceCount(&nDevices));
    if (device < 0 || device >= nDevices) return fail(ISAAC_GPU_EINVAL, "no such HIP device (this library has no CPU path)");
    HIP_CHECK(hipSetDevice(device));
    std::unique_ptr<isaac_gpu_ctx> c(new isaac_gpu_ctx);
    c->device = device; c->stream = static_cast<hipStream_t>(stream);
    if (ISAAC_GPU_STREAM_OWN == stream) { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->ownsStream = true; }
    if (const char *small = std::getenv("ISAAC_GPU_CIGAR_EXTRA_WORDS")) c->cigarExtra = std::max<u32>(1, u32(std::atol(small)));      // tests: the repeated call
    c->params = *params; c->P = makeDevParams(*params);
    if (-params->gap_open < -params->gap_extend) return fail(ISAAC_GPU_EINVAL, "gap open penalty below gap extend penalty is not supported by the banded Smith-Waterman scan");
    setAdapters(c.get());
    double tables[200]; makeQualityTables(tables, tables + 100);
    c->logTables.reserve(200);
    HIP_CHECK(hipMemcpy(c->logTables.p, tables, sizeof(tables), hipMemcpyHostToDevice));
    c->counters.reserve(COUNTER_SHARDS);
    HIP_CHECK(hipMemset(c->counters.p, 0, COUNTER_SHARDS * sizeof(Counters)));
    c->overflowCount.reserve(1);
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cluster_sums_xl), hipFuncAttributeMaxDynamicSharedMemorySize, int(SUMS_XL_LDS)));
    if (const char *e = getenv("ISAAC_GPU_CHUNK_CLUSTERS")) c->chunkClusters = u32(std::max(1024, atoi(e)));
    *out = c.release();
    return ISAAC_GPU_OK;
    ISAAC_CATCH
}

void isaac_gpu_destroy(isaac_gpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    resolveTimers(c);
#if defined(ISAAC_KERNEL_STAMPS)
    {
        hipDeviceSynchronize();
        for (void (*print)() : stampPrinters()) print();
    }
#endif
    for (hipEvent_t e : c->eventPool) hipEventDestroy(e);
    if (c->downloadStream) { hipStreamSynchronize(c->downloadStream); for (auto &d : c->downloads) hipEventDestroy(d.second); hipStreamDestroy(c->downloadStream); }
    if (c->bswStream) { hipStreamSynchronize(c->bswStream); hipStreamDestroy(c->bswStream); hipEventDestroy(c->bswBegin); hipEventDestroy(c->bswEnd); }
    if (c->ownsStream) hipStreamDestroy(c->stream);
    for (isaac_host_resolve::Resolver *r : c->resolvers) isaac_host_resolve::destroy(r);
    delete c;
}

int isaac_gpu_malloc(isaac_gpu_ctx *c, uint64_t bytes, void **dev) { ISAAC_TRY HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMalloc(dev, bytes ? bytes : 16)); return 0; ISAAC_CATCH }
int isaac_gpu_free(isaac_gpu_ctx *c, void *dev) { ISAAC_TRY HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipFree(dev)); return 0; ISAAC_CATCH }
int isaac_gpu_upload(isaac_gpu_ctx *c, void *dev, const void *host, uint64_t bytes)
{ ISAAC_TRY HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream)); HIP_CHECK(hipStreamSynchronize(c->stream)); return 0; ISAAC_CATCH }
int isaac_gpu_download(isaac_gpu_ctx *c, void *host, const void *dev, uint64_t bytes)
{ ISAAC_TRY HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream)); HIP_CHECK(hipStreamSynchronize(c->stream)); return 0; ISAAC_CATCH }
int isaac_gpu_download_async(isaac_gpu_ctx *c, void *host, const void *dev, uint64_t bytes, uint64_t *ticketOut)
{
    ISAAC_TRY
    if (!ticketOut) return fail(ISAAC_GPU_EINVAL, "null argument");
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->downloadStream) HIP_CHECK(hipStreamCreateWithFlags(&c->downloadStream, hipStreamNonBlocking));
    hipEvent_t ready, done;
    HIP_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ready, c->stream));                            // what the context's stream has been given so far
    HIP_CHECK(hipStreamWaitEvent(c->downloadStream, ready, 0));
    if (bytes) HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->downloadStream));
    HIP_CHECK(hipEventRecord(done, c->downloadStream));
    HIP_CHECK(hipEventDestroy(ready));                                      // (destroyed when the work it marks has passed)
    c->downloads.push_back(std::make_pair(++c->downloadTicket, done));
    *ticketOut = c->downloadTicket;
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_download_wait(isaac_gpu_ctx *c, uint64_t ticket)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    while (!c->downloads.empty() && c->downloads.front().first <= ticket)
    {
        const hipEvent_t done = c->downloads.front().second;
        c->downloads.pop_front();
        const hipError_t e = hipEventSynchronize(done);
        hipEventDestroy(done);
        HIP_CHECK(e);
    }
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_host_malloc(uint64_t bytes, void **hostOut)
{
    ISAAC_TRY
    if (!hostOut) return fail(ISAAC_GPU_EINVAL, "null argument");
    HIP_CHECK(hipHostMalloc(hostOut, std::max<uint64_t>(bytes, 64), hipHostMallocDefault));
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_host_free(void *host) { ISAAC_TRY if (host) HIP_CHECK(hipHostFree(host)); return 0; ISAAC_CATCH }
int isaac_gpu_memory_info(isaac_gpu_ctx *c, uint64_t *freeOut, uint64_t *totalOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    if (freeOut) *freeOut = f;
    if (totalOut) *totalOut = t;
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_copy(isaac_gpu_ctx *c, void *dstDev, const void *srcDev, uint64_t bytes)
{ ISAAC_TRY HIP_CHECK(hipSetDevice(c->device)); if (bytes) HIP_CHECK(hipMemcpyAsync(dstDev, srcDev, bytes, hipMemcpyDeviceToDevice, c->stream)); return 0; ISAAC_CATCH }
// the candidate pool of a selection was sized from a match count that was too small (isaac_gpu_select_n): clusters past its end are flagged
// and counted, and the first wait after the call says so
static int checkPoolShort(isaac_gpu_ctx *c)
{
    if (!c->poolShort.p) return 0;
    u32 flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, c->poolShort.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!flag) return 0;
    HIP_CHECK(hipMemsetAsync(c->poolShort.p, 0, 4, c->stream));
    if (flag & 2)
    {   // (with deferred completion the call cannot be repeated here: the next ones get the larger arena)
        c->cigarExtra = std::min<u32>(c->cigarExtra * 4, 2048);
        if (!(flag & 1)) return fail(ISAAC_GPU_ECAPACITY, "the CIGAR arena of a selection was used up (many single-indel or gapped alignments): some clusters are flagged (overflow_clusters); repeat the call");
    }
    return fail(ISAAC_GPU_ECAPACITY, "n_matches given to isaac_gpu_select_n is smaller than the number of matches under cluster_offsets_dev: clusters beyond it have no candidates (overflow_clusters)");
}
// ISAAC_GPU_DEBUG_TIERS=1: the last chunk's cluster counts per tier of the probability sums, on stderr (measurement aid)
static void debugTiers(isaac_gpu_ctx *c)
{
    static const bool on = std::getenv("ISAAC_GPU_DEBUG_TIERS") != nullptr;
    if (!on || !c->heavyCount.p) return;
    u32 n[16] = { 0 };
    HIP_CHECK(hipMemcpy(n, c->heavyCount.p, sizeof(n), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "isaac_gpu tiers (last chunk): residual %u, lists > 16: %u, > 64: %u, > 256: %u, > 1024: %u, > 3584: %u\n", n[0], n[4], n[8], n[1], n[3], n[2]);
}
int isaac_gpu_synchronize(isaac_gpu_ctx *c) { ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipStreamSynchronize(c->stream)); debugTiers(c); return checkPoolShort(c); ISAAC_CATCH }
int isaac_gpu_set_deferred_completion(isaac_gpu_ctx *c, int enabled)
{
    ISAAC_TRY
    if (!enabled) HIP_CHECK(hipStreamSynchronize(c->stream));
    c->deferredCompletion = enabled != 0;
    return 0;
    ISAAC_CATCH
}

// 32 bases per thread: two words of 2-bit codes and one word of not-ACGT flags
__global__ void k_pack_reference(const char *bases, u64 totalBases, u32 *packed, u32 *notBase, u64 nWords32)
{
    const u64 w = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= nWords32) return;
    u32 lo = 0, hi = 0, bad = 0;
    for (u32 i = 0; i < 32; ++i)
    {
        const u64 at = w * 32 + i;
        const u32 ch = at < totalBases ? u32(u8(bases[at])) : u32('N');
        const u32 d = ch - 0x41u;
        const bool isBase = d < 32u && ((0x80045u >> d) & 1u);           // A C G T, upper case
        const u32 code = isBase ? ((ch >> 1) & 3u) : 0u;
        if (i < 16) lo |= code << (2 * i); else hi |= code << (2 * (i - 16));
        bad |= (isBase ? 0u : 1u) << i;
    }
    packed[2 * w] = lo; packed[2 * w + 1] = hi; notBase[w] = bad;
}

// what isaac_gpu_resolve_flagged keeps between calls is made from the parameters, the contigs and the loaded-contig flags: whoever changes one of them drops it
static void dropResolvers(isaac_gpu_ctx *c, bool basesToo)
{
    for (isaac_host_resolve::Resolver *r : c->resolvers) isaac_host_resolve::destroy(r);
    c->resolvers.clear(); c->resolverLoaded.clear();
    if (basesToo) { c->hostBases.clear(); c->hostBases.shrink_to_fit(); }
}

static void setContigs(isaac_gpu_ctx *c, const uint64_t *offsets, uint32_t n, bool pack = true)
{
    dropResolvers(c, true);
    c->hostBasesGiven = nullptr;                  // (the caller's copy was of the contigs before these)
    c->nContigs = n; c->hContigOffset.assign(offsets, offsets + n + 1);
    c->contigOffset.reserve(n + 1);
    HIP_CHECK(hipMemcpy(c->contigOffset.p, offsets, (n + 1) * sizeof(u64), hipMemcpyHostToDevice));
    c->hContigLoaded.assign(n, 1); c->contigLoaded.reserve(n);
    HIP_CHECK(hipMemcpy(c->contigLoaded.p, c->hContigLoaded.data(), n, hipMemcpyHostToDevice));
    c->contigHits.reserve(n);
    c->packedBorrowed = c->notBaseBorrowed = nullptr;
    if (!pack) return;
    const u64 nWords32 = (offsets[n] + 31) / 32 + 4;                       // + spare words: a lane reads two words past its first
    c->packedBases.reserve(2 * nWords32); c->notBase.reserve(nWords32);
    k_pack_reference<<<gridFor(nWords32, 256), 256, 0, c->stream>>>(c->bases, offsets[n], c->packedBases.p, c->notBase.p, nWords32);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

int isaac_gpu_load_contigs(isaac_gpu_ctx *c, const char *bases, const uint64_t *offsets, uint32_t n)
{
    ISAAC_TRY
    if (!n || !bases || !offsets) return fail(ISAAC_GPU_EINVAL, "no contigs");
    HIP_CHECK(hipSetDevice(c->device));
    c->basesOwned.reserve(offsets[n] + 64);
    HIP_CHECK(hipMemset(c->basesOwned.p, 'N', offsets[n] + 64));
    HIP_CHECK(hipMemcpy(c->basesOwned.p, bases, offsets[n], hipMemcpyHostToDevice));
    c->bases = c->basesOwned.p;
    setContigs(c, offsets, n);
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_load_contigs_dev(isaac_gpu_ctx *c, const char *bases_dev, const uint64_t *offsets, uint32_t n)
{
    ISAAC_TRY
    if (!n || !bases_dev || !offsets) return fail(ISAAC_GPU_EINVAL, "no contigs");
    HIP_CHECK(hipSetDevice(c->device));
    c->bases = bases_dev;
    setContigs(c, offsets, n);
    return 0;
    ISAAC_CATCH
}

// The mask files are streamed in mask order through two device staging buffers (the copy of one chunk overlaps the split of the
// previous one); the host never holds more than the caller's own mappings.
int isaac_gpu_load_index(isaac_gpu_ctx *c, const isaac_reference_kmer *const *masks, const uint64_t *sizes, uint32_t nMasks, const uint32_t *karyotype, uint32_t nContigs)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nMasks && (!masks || !sizes)) return fail(ISAAC_GPU_EINVAL, "masks_host and mask_sizes are required");
    if (karyotype && nContigs != c->nContigs) return fail(ISAAC_GPU_EINVAL, "karyotype_of_contig needs one entry per loaded contig");
    u64 total = 0; for (u32 m = 0; m < nMasks; ++m) total += sizes[m];
    if (total >= (u64(1) << 32)) return fail(ISAAC_GPU_EINVAL, "tables of 2^32 entries and more are not supported");
    hipStream_t st = c->stream;
    c->entriesBorrowed = nullptr;
    c->entries.reserve(total + 4); c->nKmers = 0; c->prefixBits = 0;             // + 4: k_find_matches reads four entries from the first candidate on
    // The mask files arrive as host memory (memory-mapped files, usually): 47 GB for GRCh38.  Copied by the runtime straight from pageable memory they
    // move at 3-4 GB/s (one thread faulting the pages in and staging them).  Here a few host threads copy 256 MB pieces into a ring of pinned buffers side by
    // side -- that is where the page faults and the page-cache reads happen -- and the pieces go from there straight to their place in the table (its
    // entries are the files' records; rounds 2-4 kept k-mers and positions in two arrays and split every piece on the device).
    const u64 chunk = 1u << 24;      // records per staging buffer (256 MB)
    struct Piece { const ReferenceKmerRecord *src; u64 n, at; };
    std::vector<Piece> pieces;
    c->maskOffsets.assign(1, 0);
    {
        u64 at = 0;
        for (u32 m = 0; m < nMasks; ++m)
        {
            for (u64 done = 0; done < sizes[m]; done += chunk) { const u64 n = std::min(chunk, sizes[m] - done); pieces.push_back(Piece{ reinterpret_cast<const ReferenceKmerRecord *>(masks[m]) + done, n, at }); at += n; }
            c->maskOffsets.push_back(at);
        }
    }
    const u32 SLOTS = 6;
    // (ISAAC_GPU_LOAD_THREADS: measurement aid)
    const u32 FILLERS = std::getenv("ISAAC_GPU_LOAD_THREADS") ? std::max(1, std::atoi(std::getenv("ISAAC_GPU_LOAD_THREADS"))) : 8;
    DevBuf<u32> disorder; disorder.reserve(1);
    HIP_CHECK(hipMemsetAsync(disorder.p, 0, 4, st));
    // (ISAAC_GPU_LOAD_STREAMS=2: the pieces on two copy streams in turn -- measured no faster than one, 1.77 against 1.60 s for 47 GB, nor were 12 or 16 copying
    // threads: 28 GB/s is what this host's link gives one direction, profiles/r5_l_cli_timing.log)
    const u32 nCopyStreams = std::getenv("ISAAC_GPU_LOAD_STREAMS") && 2 == std::atoi(std::getenv("ISAAC_GPU_LOAD_STREAMS")) ? 2u : 1u;
    hipStream_t copyStreams[2] = { nullptr, nullptr };
    for (u32 i = 0; i < nCopyStreams; ++i) HIP_CHECK(hipStreamCreateWithFlags(&copyStreams[i], hipStreamNonBlocking));
    hipEvent_t copied[2], split[2], left[SLOTS];
    for (u32 i = 0; i < 2; ++i) { HIP_CHECK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&split[i], hipEventDisableTiming)); }
    for (u32 i = 0; i < SLOTS; ++i) HIP_CHECK(hipEventCreateWithFlags(&left[i], hipEventDisableTiming));
    ReferenceKmerRecord *pinned[SLOTS] = { nullptr };
    std::vector<std::thread> fillers;
    std::vector<std::atomic<int> > filled(pieces.size());
    for (auto &f : filled) f.store(0);
    std::atomic<size_t> nextPiece(0), consumed(0);      // pieces handed to a filler / pieces whose slot may be written again
    std::atomic<bool> stop(false);
    auto cleanup = [&]()
    {
        stop = true;
        for (std::thread &t : fillers) t.join();
        for (u32 i = 0; i < nCopyStreams; ++i) { hipStreamSynchronize(copyStreams[i]); hipStreamDestroy(copyStreams[i]); }
        for (u32 i = 0; i < 2; ++i) { hipEventDestroy(copied[i]); hipEventDestroy(split[i]); }
        for (u32 i = 0; i < SLOTS; ++i) { hipEventDestroy(left[i]); if (pinned[i]) hipHostFree(pinned[i]); }
    };
    try
    {
        HIP_CHECK(hipStreamSynchronize(st));
        if (!pieces.empty())
        {
            for (u32 i = 0; i < SLOTS; ++i) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&pinned[i]), chunk * sizeof(ReferenceKmerRecord), hipHostMallocDefault));
            for (u32 t = 0; t < FILLERS; ++t)
                fillers.emplace_back([&]()
                {
                    for (size_t k = nextPiece++; k < pieces.size() && !stop; k = nextPiece++)
                    {
                        while (k >= consumed + SLOTS && !stop) std::this_thread::yield();      // the slot's last piece has not left for the device yet
                        if (stop) return;
                        std::memcpy(pinned[k % SLOTS], pieces[k].src, pieces[k].n * sizeof(ReferenceKmerRecord));
                        filled[k].store(1, std::memory_order_release);
                    }
                });
            bool used[2] = { false, false };
            for (size_t k = 0; k < pieces.size(); ++k)
            {
                const u32 turn = u32(k & 1), slot = u32(k % SLOTS);
                hipStream_t copyStream = copyStreams[turn % nCopyStreams];
                while (!filled[k].load(std::memory_order_acquire)) std::this_thread::yield();
                HIP_CHECK(hipMemcpyAsync(c->entries.p + pieces[k].at, pinned[slot], pieces[k].n * sizeof(ReferenceKmerRecord), hipMemcpyHostToDevice, copyStream));
                HIP_CHECK(hipEventRecord(copied[turn], copyStream));
                HIP_CHECK(hipEventRecord(left[slot], copyStream));
                HIP_CHECK(hipStreamWaitEvent(st, copied[turn], 0));
                k_check_order<<<gridFor(pieces[k].n, 256), 256, 0, st>>>(c->entries.p, pieces[k].at, pieces[k].n, disorder.p);     // (pieces land in order: the entry before a piece is there)
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipEventRecord(split[turn], st)); used[turn] = true;
                // two copies may be queued; the slots of everything before them are free again
                if (k >= 2) { HIP_CHECK(hipEventSynchronize(left[(k - 2) % SLOTS])); consumed = k - 1; }
            }
        }
        u32 bad = 0;
        HIP_CHECK(hipMemcpyAsync(&bad, disorder.p, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st)); for (u32 i = 0; i < nCopyStreams; ++i) HIP_CHECK(hipStreamSynchronize(copyStreams[i]));
        cleanup();
        if (bad) return fail(ISAAC_GPU_EINVAL, "mask files are not in global k-mer order");
    }
    catch (...) { cleanup(); throw; }
    c->nKmers = total;
    buildPrefixTable(c);
    c->hasKaryotype = false;
    if (karyotype)
    {
        bool identity = true; for (u32 i = 0; i < nContigs; ++i) identity &= karyotype[i] == i;
        if (!identity) { c->karyotype.reserve(nContigs); HIP_CHECK(hipMemcpy(c->karyotype.p, karyotype, nContigs * 4, hipMemcpyHostToDevice)); c->hasKaryotype = true; }
    }
    return 0;
    ISAAC_CATCH
}

namespace
{
// One mask of the table (see index_kernels.h): its k-mers emitted in position order, sorted, analysed and appended.
struct IndexScratch
{
    DevBuf<u64> keys0, vals0, keys, vals, blockBase;
    DevBuf<u32> head, isFwd, runId, fwdExcl, runStart, runTotal, runFwd, emit, emitSlot;
    void release()
    {
        keys0.release(); vals0.release(); keys.release(); vals.release(); blockBase.release(); head.release(); isFwd.release(); runId.release(); fwdExcl.release();
        runStart.release(); runTotal.release(); runFwd.release(); emit.release(); emitSlot.release();
    }
};

// out byte j of the shuffled key = key byte src[j] (byte 0 = least significant)
ByteShuffle makeShuffle(const u32 src[8])
{
    ByteShuffle s; s.selLo = 0; s.selHi = 0;
    for (u32 j = 0; j < 4; ++j) { s.selLo |= src[j] << (8 * j); s.selHi |= src[4 + j] << (8 * j); }
    return s;
}
// layout of the key under a choice of 4 of the 8 four-base blocks: the chosen bytes in the low half (the sorted half, see
// k_mark_neighbors), the others above, both in descending significance.  at[b] = where byte b of the k-mer sits.
void blockLayout(u32 chosen, u32 at[8])
{
    u32 hi = 7, lo = 3;
    for (int b = 7; b >= 0; --b) { if ((chosen >> b) & 1) at[b] = lo--; else at[b] = hi--; }
}
} // namespace

int isaac_gpu_build_index(isaac_gpu_ctx *c, uint32_t repeatThreshold, int annotateNeighbors, uint64_t *nEntriesOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->bases) return fail(ISAAC_GPU_EINVAL, "load the contigs first");
    const u64 totalBases = c->hContigOffset[c->nContigs];
    hipStream_t st = c->stream;
    const u32 *packed = c->packedWords(), *notBase = c->notBaseWords();
    const u64 nBlocks = (totalBases + INDEX_TILE - 1) / INDEX_TILE;
    if (nBlocks >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "reference too long");
    c->entriesBorrowed = nullptr;
    c->nKmers = 0; c->prefixBits = 0; c->hasKaryotype = false; c->maskOffsets.assign(1, 0);
    // 1. how many k-mers of each mask every block of positions holds
    DevBuf<u32> counts; counts.reserve(size_t(INDEX_MASKS) * nBlocks);
    DevBuf<unsigned long long> validCount; validCount.reserve(1);
    HIP_CHECK(hipMemsetAsync(validCount.p, 0, 8, st));
    k_index_count<<<u32(nBlocks), INDEX_THREADS, 0, st>>>(packed, notBase, c->contigOffset.p, c->nContigs, totalBases, nBlocks, counts.p, validCount.p);
    HIP_CHECK(hipGetLastError());
    unsigned long long nValid = 0;
    HIP_CHECK(hipMemcpyAsync(&nValid, validCount.p, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (nValid >= (u64(1) << 32)) return fail(ISAAC_GPU_EINVAL, "tables of 2^32 entries and more are not supported");
    // every stored entry is a forward-strand occurrence: nValid bounds the table, 2 * nValid the distinct k-mers of both strands
    c->entries.reserve(nValid + 4);
    DevBuf<u64> distinct; DevBuf<u32> entryRun;
    if (annotateNeighbors) { distinct.reserve(2 * nValid + 1); entryRun.reserve(nValid + 1); }
    std::vector<u64> distinctBase(1, 0);
    IndexScratch w;
    w.blockBase.reserve(nBlocks + 1);
    u64 nOut = 0, nDistinct = 0;
    for (u32 mask = 0; mask < INDEX_MASKS; ++mask)
    {
        // 2. the mask's k-mers in position order
        k_index_widen<<<gridFor(nBlocks, 256), 256, 0, st>>>(counts.p + size_t(mask) * nBlocks, nBlocks, w.blockBase.p);
        u64 lastCount = 0, lastBase = 0;
        HIP_CHECK(hipMemcpyAsync(&lastCount, w.blockBase.p + nBlocks - 1, 8, hipMemcpyDeviceToHost, st));
        exclusiveSum(c, w.blockBase.p, w.blockBase.p, nBlocks);
        HIP_CHECK(hipMemcpyAsync(&lastBase, w.blockBase.p + nBlocks - 1, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const u64 n = lastBase + lastCount;
        if (n >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "more than 2^31 k-mers in one mask (a degenerate reference)");
        if (n)
        {
            w.keys0.reserve(n); w.vals0.reserve(n); w.keys.reserve(n); w.vals.reserve(n);
            k_index_emit<<<u32(nBlocks), INDEX_THREADS, 0, st>>>(packed, notBase, c->contigOffset.p, c->nContigs, totalBases, mask, w.blockBase.p, w.keys0.p, w.vals0.p);
            HIP_CHECK(hipGetLastError());
            // 3. sorted by k-mer (stable: position order inside a run, forward strand first); the mask bits are equal
            sortPairs(c, w.keys0.p, w.keys.p, w.vals0.p, w.vals.p, n, 64 - int(INDEX_MASK_BITS));
            // 4. runs of equal k-mers: ReferenceSorter.cpp:179-252
            w.head.reserve(n); w.isFwd.reserve(n); w.runId.reserve(n); w.fwdExcl.reserve(n); w.emit.reserve(n); w.emitSlot.reserve(n);
            k_run_heads<<<gridFor(n, 256), 256, 0, st>>>(w.keys.p, n, w.head.p, w.isFwd.p, w.vals.p);
            inclusiveSum(c, w.head.p, w.runId.p, n);
            exclusiveSum(c, w.isFwd.p, w.fwdExcl.p, n);
            u32 nRuns = 0;
            HIP_CHECK(hipMemcpyAsync(&nRuns, w.runId.p + n - 1, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            w.runStart.reserve(nRuns); w.runTotal.reserve(nRuns); w.runFwd.reserve(nRuns);
            k_run_totals<<<gridFor(n, 256), 256, 0, st>>>(w.keys.p, n, w.runId.p, w.fwdExcl.p, w.isFwd.p, w.runStart.p, w.runTotal.p, w.runFwd.p);
            k_run_finish<<<gridFor(nRuns, 256), 256, 0, st>>>(nRuns, w.fwdExcl.p, w.runStart.p, w.runTotal.p, w.runFwd.p);
            k_emit_flags<<<gridFor(n, 256), 256, 0, st>>>(n, w.runId.p, w.isFwd.p, w.runStart.p, w.runTotal.p, w.runFwd.p, repeatThreshold, w.emit.p);
            exclusiveSum(c, w.emit.p, w.emitSlot.p, n);
            u32 lastE = 0, lastS = 0;
            HIP_CHECK(hipMemcpyAsync(&lastE, w.emit.p + n - 1, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(&lastS, w.emitSlot.p + n - 1, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            k_emit_entries<<<gridFor(n, 256), 256, 0, st>>>(w.keys.p, w.vals.p, n, w.runId.p, w.runTotal.p, repeatThreshold, w.emit.p, w.emitSlot.p, nOut, c->entries.p,
                                                              annotateNeighbors ? entryRun.p : nullptr);
            if (annotateNeighbors) k_distinct<<<gridFor(n, 256), 256, 0, st>>>(w.keys.p, n, w.head.p, w.runId.p, nDistinct, distinct.p);
            HIP_CHECK(hipGetLastError());
            nOut += u64(lastE) + lastS; nDistinct += nRuns;
        }
        c->maskOffsets.push_back(nOut); distinctBase.push_back(nDistinct);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    w.release(); counts.release();
    if (annotateNeighbors && nDistinct)
    {
        // 5. NeighborsFinder::generateNeighbors: 70 groupings of the distinct k-mers of both strands
        DevBuf<u64> keysAlt; DevBuf<u8> flags, flagsAlt;
        keysAlt.reserve(nDistinct); flags.reserve(nDistinct); flagsAlt.reserve(nDistinct);
        HIP_CHECK(hipMemsetAsync(flags.p, 0, nDistinct, st));
        rocprim::double_buffer<u64> dk(distinct.p, keysAlt.p); rocprim::double_buffer<u8> df(flags.p, flagsAlt.p);
        u32 at[8]; for (u32 b = 0; b < 8; ++b) at[b] = b;           // where byte b of the k-mer sits in the keys right now
        auto shuffleTo = [&](const u32 next[8])
        {
            u32 src[8];                                                // out byte next[b] = current byte at[b]
            for (u32 b = 0; b < 8; ++b) src[next[b]] = at[b];
            k_shuffle_keys<<<gridStrided(nDistinct, 256), 256, 0, st>>>(dk.current(), nDistinct, makeShuffle(src));
            for (u32 b = 0; b < 8; ++b) at[b] = next[b];
        };
        auto sortKeys = [&](int endBit)
        {
            size_t bytes = 0;                               // (rocPRIM takes the count as size_t: the distinct k-mers of both strands of GRCh38 are more than 2^31)
            HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, dk, df, size_t(nDistinct), 0u, unsigned(endBit), st));
            c->cubTemp.reserve(bytes + 16);
            HIP_CHECK(rocprim::radix_sort_pairs(c->cubTemp.p, bytes, dk, df, size_t(nDistinct), 0u, unsigned(endBit), st));
        };
        for (u32 chosen = 0; chosen < 256; ++chosen)
        {   // any 4 of the 8 blocks of 4 bases (oligo/Permutate.cpp:94-145, getPermutateList(4) at NeighborsFinder.cpp:196)
            if (__builtin_popcount(chosen) != 4) continue;
            u32 next[8]; blockLayout(chosen, next);
            shuffleTo(next);
            sortKeys(32);                                              // groups of equal chosen blocks
            k_mark_neighbors<<<gridStrided(nDistinct, 256), 256, 0, st>>>(dk.current(), df.current(), nDistinct);
            HIP_CHECK(hipGetLastError());
        }
        u32 identity[8]; for (u32 b = 0; b < 8; ++b) identity[b] = b;
        shuffleTo(identity);
        sortKeys(64);                                                  // back in the order of `distinct`: flags[i] belongs to distinct k-mer i
        for (u32 mask = 0; mask < INDEX_MASKS; ++mask)
        {
            const u64 first = c->maskOffsets[mask], n = c->maskOffsets[mask + 1] - first;
            if (n) k_apply_neighbors<<<gridFor(n, 256), 256, 0, st>>>(c->entries.p, entryRun.p, first, n, distinctBase[mask], df.current());
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    c->cubTemp.release();
    c->nKmers = nOut;
    buildPrefixTable(c);
    if (nEntriesOut) *nEntriesOut = nOut;
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_get_index(isaac_gpu_ctx *c, isaac_reference_kmer *out, uint64_t capacity, uint64_t *nOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nOut) *nOut = c->nKmers;
    if (!out) return 0;
    if (capacity < c->nKmers) return fail(ISAAC_GPU_ECAPACITY, "index buffer too small");
    static_assert(sizeof(ReferenceKmerRecord) == sizeof(isaac_reference_kmer), "mask file record");
    // the table's entries are the mask files' records
    if (c->nKmers) HIP_CHECK(hipMemcpyAsync(out, c->tableEntries(), c->nKmers * sizeof(ReferenceKmerRecord), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_get_index_range(isaac_gpu_ctx *c, uint64_t first, uint64_t n, isaac_reference_kmer *out)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (first > c->nKmers || n > c->nKmers - first) return fail(ISAAC_GPU_EINVAL, "range outside the table");
    if (!n) return 0;
    if (!out) return fail(ISAAC_GPU_EINVAL, "out_host is required");
    HIP_CHECK(hipMemcpyAsync(out, c->tableEntries() + first, n * sizeof(ReferenceKmerRecord), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
    ISAAC_CATCH
}

// entries of the resident table before each mask (n_masks + 1 values, mask = the k-mer's top 6 bits): where a writer of
// sorted-reference mask files cuts it
int isaac_gpu_get_mask_offsets(isaac_gpu_ctx *c, uint64_t *offsetsOut, uint32_t nMasks)
{
    ISAAC_TRY
    if (!offsetsOut || nMasks + 1 != c->maskOffsets.size()) return fail(ISAAC_GPU_EINVAL, "the resident table has a different number of masks");
    std::copy(c->maskOffsets.begin(), c->maskOffsets.end(), offsetsOut);
    return 0;
    ISAAC_CATCH
}

// the resident table as device pointers (read-only; valid until the table is rebuilt, reloaded or its owner destroyed)
int isaac_gpu_index_dev(isaac_gpu_ctx *c, const isaac_reference_kmer **entriesOut, uint64_t *nOut)
{
    ISAAC_TRY
    if (entriesOut) *entriesOut = reinterpret_cast<const isaac_reference_kmer *>(c->tableEntries());
    if (nOut) *nOut = c->nKmers;
    return 0;
    ISAAC_CATCH
}

// adopts a table that lives in the caller's device memory (another context's, or what an RCCL broadcast delivered): nothing is
// copied; only the prefix directory is built
int isaac_gpu_set_index_dev(isaac_gpu_ctx *c, const isaac_reference_kmer *entriesIn, uint64_t n, const uint64_t *maskOffsets, uint32_t nMasks)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    const TableEntry *entries = reinterpret_cast<const TableEntry *>(entriesIn);
    if (n && !entries) return fail(ISAAC_GPU_EINVAL, "entries_dev is required");
    if (n >= (u64(1) << 32)) return fail(ISAAC_GPU_EINVAL, "tables of 2^32 entries and more are not supported");
    if (maskOffsets && (maskOffsets[0] != 0 || maskOffsets[nMasks] != n)) return fail(ISAAC_GPU_EINVAL, "mask_offsets must run from 0 to n_entries");
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->entries.p && entries == c->entries.p && n == c->nKmers) return 0;      // the context's own table handed back to it: nothing to adopt (and nothing to free)
    c->entries.release();
    c->entriesBorrowed = entries; c->nKmers = n; c->hasKaryotype = false;
    if (maskOffsets) c->maskOffsets.assign(maskOffsets, maskOffsets + nMasks + 1); else { c->maskOffsets.assign(1, 0); c->maskOffsets.push_back(n); }
    buildPrefixTable(c);
    return 0;
    ISAAC_CATCH
}

// the table of another context for this one: in place when both are on one device, a copy when they are not; the karyotype translation with it
int isaac_gpu_share_index(isaac_gpu_ctx *c, isaac_gpu_ctx *owner)
{
    ISAAC_TRY
    if (!c || !owner || c == owner) return fail(ISAAC_GPU_EINVAL, "two different contexts are required");
    if (c->nContigs != owner->nContigs) return fail(ISAAC_GPU_EINVAL, "both contexts must have loaded the same contigs");
    HIP_CHECK(hipSetDevice(owner->device));
    HIP_CHECK(hipStreamSynchronize(owner->stream));
    const u64 n = owner->nKmers;
    const TableEntry *entries = owner->tableEntries();
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->entries.release();
    // (ISAAC_GPU_SHARE_BY_COPY: tests on a box with one device take the branch two devices take)
    if ((c->device != owner->device || std::getenv("ISAAC_GPU_SHARE_BY_COPY")) && n)
    {
        c->entries.reserve(n + 4);
        HIP_CHECK(hipMemcpy(c->entries.p, entries, n * sizeof(TableEntry), hipMemcpyDefault));
        c->entriesBorrowed = nullptr;
    }
    else c->entriesBorrowed = entries;
    c->nKmers = n; c->maskOffsets = owner->maskOffsets;
    c->hasKaryotype = owner->hasKaryotype;
    if (owner->hasKaryotype)
    {
        std::vector<u32> k(owner->nContigs);
        HIP_CHECK(hipSetDevice(owner->device));
        HIP_CHECK(hipMemcpy(k.data(), owner->karyotype.p, k.size() * 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipSetDevice(c->device));
        c->karyotype.reserve(k.size());
        HIP_CHECK(hipMemcpy(c->karyotype.p, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    }
    buildPrefixTable(c);
    return 0;
    ISAAC_CATCH
}

// contigs and table of another context for this one, which need not have loaded anything: on one device everything is read where the owner has it (bases, their
// packed copy, table, prefix directory: no byte copied, none computed -- the owner must outlive this context or load something else first); on another
// device the bases and the table are copied over the link and the rest is made from them.  The resolved-on-host copy of the contigs (isaac_gpu_set_host_contigs)
// goes along.
int isaac_gpu_share_reference(isaac_gpu_ctx *c, isaac_gpu_ctx *owner)
{
    ISAAC_TRY
    if (!c || !owner || c == owner) return fail(ISAAC_GPU_EINVAL, "two different contexts are required");
    if (!owner->bases || !owner->nContigs) return fail(ISAAC_GPU_EINVAL, "the owner has no contigs");
    const bool inPlace = c->device == owner->device && !std::getenv("ISAAC_GPU_SHARE_BY_COPY");
    HIP_CHECK(hipSetDevice(owner->device));
    HIP_CHECK(hipStreamSynchronize(owner->stream));
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    const u64 totalBases = owner->hContigOffset[owner->nContigs];
    if (inPlace)
    {
        c->basesOwned.release(); c->packedBases.release(); c->notBase.release();
        c->bases = owner->bases;
        setContigs(c, owner->hContigOffset.data(), owner->nContigs, false);
        c->packedBorrowed = owner->packedWords(); c->notBaseBorrowed = owner->notBaseWords();
    }
    else
    {
        c->basesOwned.reserve(totalBases + 64);
        HIP_CHECK(hipMemset(c->basesOwned.p, 'N', totalBases + 64));
        HIP_CHECK(hipMemcpy(c->basesOwned.p, owner->bases, totalBases, hipMemcpyDefault));
        c->bases = c->basesOwned.p;
        setContigs(c, owner->hContigOffset.data(), owner->nContigs);
    }
    c->hostBasesGiven = owner->hostBasesGiven;
    if (!owner->tableEntries()) return 0;
    if (!inPlace) return isaac_gpu_share_index(c, owner);
    c->entries.release(); c->prefixTable.release();
    c->entriesBorrowed = owner->tableEntries(); c->nKmers = owner->nKmers; c->maskOffsets = owner->maskOffsets;
    c->hasKaryotype = owner->hasKaryotype;
    if (owner->hasKaryotype)
    {
        c->karyotype.reserve(owner->nContigs);
        HIP_CHECK(hipMemcpy(c->karyotype.p, owner->karyotype.p, size_t(owner->nContigs) * 4, hipMemcpyDeviceToDevice));
    }
    c->prefixBits = owner->prefixBits; c->prefixBorrowed = owner->prefixBits ? owner->prefixWords() : nullptr;
    return 0;
    ISAAC_CATCH
}

// other options / read geometry for the same reference and table (the reference constructs its MatchFinder / MatchSelector once per
// run; a service that aligns runs of different read lengths against one resident genome does not reload 50 GB for that)
int isaac_gpu_set_params(isaac_gpu_ctx *c, const isaac_params *params)
{
    ISAAC_TRY
    if (!params) return fail(ISAAC_GPU_EINVAL, "null argument");
    if (-params->gap_open < -params->gap_extend) return fail(ISAAC_GPU_EINVAL, "gap open penalty below gap extend penalty is not supported by the banded Smith-Waterman scan");
    const DevParams P = makeDevParams(*params);
    (void)makeDevAdapters(*params);                // (throws for a bad adapter before anything is changed)
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->params = *params; c->P = P;
    setAdapters(c);
    dropResolvers(c, false);                      // they hold the parameters they were made with
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_find_matches(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, isaac_match *matchesOut, uint64_t capacity,
                           uint64_t *clusterOffsets, uint64_t *nMatchesOut, uint8_t *contigHasMatches)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (!c->tableEntries() || !c->bases) return fail(ISAAC_GPU_EINVAL, "load the contigs and the index first");
    if (!matchesOut || !clusterOffsets) return fail(ISAAC_GPU_EINVAL, "matches_dev and cluster_offsets_dev are required");
    if (nClusters > 0x7fffffffu || tile > 0xfff) return fail(ISAAC_GPU_EINVAL, "SeedId overflow (SeedId.hh:95-109)");
    hipStream_t st = c->stream;
    const DevParams &P = c->P;
    const u32 stride = 2 * P.nSeeds * std::max(1u, P.repeatThreshold - 1);
    const u32 chunk = chunkFor(c, nClusters);
    c->staging.reserve(size_t(chunk) * stride); c->counts.reserve(chunk); c->chunkOffsets.reserve(chunk);
    HIP_CHECK(hipMemsetAsync(c->contigHits.p, 0, c->nContigs * 4, st));
    const DevReference R = c->ref();
    c->matchBase.reserve(1);
    HIP_CHECK(hipMemsetAsync(c->matchBase.p, 0, 8, st));
    for (u32 done = 0; done < nClusters; done += chunk)
    {
        const u32 n = std::min(chunk, nClusters - done);
        {
            ScopedTimer t(c, "find_matches");
            const size_t lds = size_t(FIND_CLUSTERS_PER_BLOCK) * P.clusterLength + 16;
            k_find_matches<<<gridFor(n, FIND_CLUSTERS_PER_BLOCK), FIND_BLOCK, lds, st>>>(P, R, bcl + u64(done) * P.clusterLength, n, done, tile,
                                                                                          c->staging.p, c->counts.p, stride, c->contigHits.p, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        ScopedTimer t(c, "compact_matches");
        exclusiveSum(c, c->counts.p, c->chunkOffsets.p, n);
        k_compact_matches<<<gridFor(n, 256), 256, 0, st>>>(c->staging.p, c->counts.p, c->chunkOffsets.p, n, stride, c->matchBase.p,
                                                            reinterpret_cast<Match *>(matchesOut), capacity, clusterOffsets + done);
        k_advance_match_base<<<1, 1, 0, st>>>(c->matchBase.p, c->chunkOffsets.p, c->counts.p, n, clusterOffsets + done + n);
        HIP_CHECK(hipGetLastError());
    }
    u64 base = 0;
    if (!nClusters) HIP_CHECK(hipMemcpyAsync(clusterOffsets, &base, 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(&base, c->matchBase.p, 8, hipMemcpyDeviceToHost, st));
    if (contigHasMatches) { c->hContigHits.resize(c->nContigs); HIP_CHECK(hipMemcpyAsync(c->hContigHits.data(), c->contigHits.p, c->nContigs * 4, hipMemcpyDeviceToHost, st)); }
    HIP_CHECK(hipStreamSynchronize(st));      // the one host wait of the call: the match count and the hit flags
    if (contigHasMatches) for (u32 i = 0; i < c->nContigs; ++i) contigHasMatches[i] |= u8(c->hContigHits[i] != 0);
    if (nMatchesOut) *nMatchesOut = base;
    if (base > capacity) return fail(ISAAC_GPU_ECAPACITY, "matches_dev is too small");
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_set_loaded_contigs(isaac_gpu_ctx *c, const uint8_t *loaded, uint32_t n)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (n != c->nContigs) return fail(ISAAC_GPU_EINVAL, "contig count mismatch");
    if (loaded) c->hContigLoaded.assign(loaded, loaded + n); else c->hContigLoaded.assign(n, 1);
    HIP_CHECK(hipMemcpy(c->contigLoaded.p, c->hContigLoaded.data(), n, hipMemcpyHostToDevice));
    return 0;
    ISAAC_CATCH
}

static GappedBuffers gappedBuffers(isaac_gpu_ctx *c, u32 which)
{
    GappedBuffers gb;
    gb.cap = 2 * c->chunkNow;
    // the rescue stage has its own arrays (its results are read by k_select and the wave-per-cluster pass at the end of the chunk)
    DevBuf<GappedJob> &jobs = which ? c->rescueGappedJobs : c->gappedJobs; DevBuf<GappedResult> &results = which ? c->rescueGappedResults : c->gappedResults;
    jobs.reserve(gb.cap); results.reserve(gb.cap); c->gappedBase.reserve(c->chunkNow); c->gappedCounters.reserve(4);
    gb.jobs = jobs.p; gb.results = results.p; gb.base = c->gappedBase.p; gb.counter = c->gappedCounters.p + which;
    return gb;
}

// timer: k_gapped_jobs alone (so that the figure can be set against that kernel's duration in a rocprofv3 trace); rescanTimer: k_gapped_rescan
static void launchGappedJobs(isaac_gpu_ctx *c, const uint8_t *bcl, u32 clusterBase, const GappedBuffers &gb, const char *timer, const char *rescanTimer)
{
    const u32 maxReadLength = std::max(c->P.readLength[0], c->P.nReads > 1 ? c->P.readLength[1] : 0u);
    // one wavefront (eight problems) per workgroup: the LDS of a workgroup is what limits the problems in flight, and at 2 x 250 five
    // workgroups of eight fit a CU where two of sixteen did
    // (2 x 250 with 5 % indel reads: 15.5 -> 13.0 ms per 1 M pairs; 2 x 150: 4.46 -> 4.44)
    // (round 6) the sequences of a problem stay in the registers of its eight lanes when they fit -- reads of up to 177 bases in three registers per lane and
    // sequence, of up to 305 in five -- and the traceback flags take ten bytes per row: 1.6 KB of LDS per problem at 2 x 150 where round 5 had 2.3 KB,
    // three wavefronts per SIMD instead of two
    const bool staged = maxReadLength > BSW_REGISTER_BASES_LONG;
    size_t lds = size_t(BSW_BLOCK / BSW_GROUP_LANES) * gappedGroupLdsBytes(maxReadLength, staged);
    // (measurements: ISAAC_GPU_GAPPED_LDS=<bytes> asks for more LDS per workgroup than the kernel uses, i.e. fewer of its wavefronts per CU and room for other contexts' kernels beside them)
    static const size_t ldsAtLeast = std::getenv("ISAAC_GPU_GAPPED_LDS") ? size_t(std::atol(std::getenv("ISAAC_GPU_GAPPED_LDS"))) : 0;
    lds = std::max(lds, std::min<size_t>(ldsAtLeast, 65536));
    {
        ScopedTimer t(c, timer);
        if (ISAAC_BSW_GLOBAL_FLAGS) c->bswFlags.reserve(size_t(65536) * (BSW_BLOCK / BSW_GROUP_LANES) * bswFlagBytes(maxReadLength));
        const auto kernel = maxReadLength <= BSW_REGISTER_BASES_SHORT ? k_gapped_jobs : staged ? k_gapped_jobs_staged : k_gapped_jobs_long;
        static const u32 grid = std::getenv("ISAAC_GPU_GAPPED_GRID") ? u32(std::max(256, std::atoi(std::getenv("ISAAC_GPU_GAPPED_GRID")))) : GAPPED_GRID;      // (measurements: workgroups the problems are dealt to)
        // (measurements: ISAAC_GPU_BSW_SIDE_STREAM=1 runs the kernel on a stream of the device's lowest priority, between two events: with several contexts on one GPU
        // the other contexts' short kernels are then dispatched ahead of its workgroups)
        static const bool sideStream = std::getenv("ISAAC_GPU_BSW_SIDE_STREAM") && std::atoi(std::getenv("ISAAC_GPU_BSW_SIDE_STREAM"));
        hipStream_t st = c->stream;
        if (sideStream)
        {
            if (!c->bswStream)
            {
                int least = 0, greatest = 0;
                HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                HIP_CHECK(hipStreamCreateWithPriority(&c->bswStream, hipStreamNonBlocking, least));
                HIP_CHECK(hipEventCreateWithFlags(&c->bswBegin, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&c->bswEnd, hipEventDisableTiming));
            }
            HIP_CHECK(hipEventRecord(c->bswBegin, c->stream)); HIP_CHECK(hipStreamWaitEvent(c->bswStream, c->bswBegin, 0));
            st = c->bswStream;
        }
        kernel<<<grid, BSW_BLOCK, lds, st>>>(c->P, c->ref(), bcl, clusterBase, gb.jobs, gb.counter, gb.cap, maxReadLength, gb.results, c->bswFlags.p);
        if (sideStream) { HIP_CHECK(hipEventRecord(c->bswEnd, c->bswStream)); HIP_CHECK(hipStreamWaitEvent(c->stream, c->bswEnd, 0)); }
    }
    HIP_CHECK(hipGetLastError());
    ScopedTimer t(c, rescanTimer);
    k_gapped_rescan<<<1024, 256, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, gb.jobs, gb.counter, gb.cap, gb.results);
    HIP_CHECK(hipGetLastError());
}

// the calls that run the fragment stage alone write the first of the two ClusterFragments buffers, sized for the chunk in use
// The candidate pools of the chunk in use (types.h: ClusterMeta, candidate pool, cigar arena).  `slots` bounds the seed matches (or
// caller-supplied candidates) of any chunk of the call; 0 = not known, the hard bound of seeds x strands x (repeat threshold - 1)
// per cluster is taken then (4.6 KB per cluster at 2x150 -- still a fifth of the fixed records this replaced).
static void preparePools(isaac_gpu_ctx *c, u64 slots)
{
    const u64 perCluster = 2 * u64(c->P.nSeeds) * std::max(1u, c->P.repeatThreshold - 1);
    const u64 hard = u64(c->chunkNow) * perCluster;
    const u64 cap = std::max<u64>(slots ? slots : hard, 64);
    // the cigar arena: three words per candidate slot and, for single-indel and accepted gapped alignments, cigarExtra words per cluster on average (32 to
    // begin with; a call whose chunks use them up is repeated with four times as many: selectFromSource)
    const u64 arena = 3 * cap + u64(c->cigarExtra) * u64(c->chunkNow);
    if (arena >= (u64(1) << 32)) throw std::invalid_argument("too many candidates in one chunk");
    c->clusterMeta.reserve(c->chunkNow); c->candPool.reserve(cap); c->cigarArena.reserve(arena); c->cigarNext.reserve(1);
    c->pools.meta = c->clusterMeta.p; c->pools.cands = c->candPool.p; c->pools.cigars = c->cigarArena.p; c->pools.candCap = u32(cap);
    c->pools.cigarCap = u32(arena); c->pools.cigarNext = c->cigarNext.p;
    if (c->P.adapters) { c->adapterRanges.reserve(4 * c->candPool.n); c->P.adapterRanges = c->adapterRanges.p; c->P.adapterCandBase = c->candPool.p; }
    if (!c->poolShort.p) { c->poolShort.reserve(1); HIP_CHECK(hipMemsetAsync(c->poolShort.p, 0, 4, c->stream)); }
    c->pools.shortFlag = c->poolShort.p;
}
// entries [0] and [n] of a device array of offsets (a host wait: only the calls outside the timed path use it)
static u64 offsetsSpan(isaac_gpu_ctx *c, const uint64_t *offsetsDev, u32 n)
{
    u64 ends[2] = { 0, 0 };
    HIP_CHECK(hipMemcpyAsync(&ends[0], offsetsDev, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(&ends[1], offsetsDev + n, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    return ends[1] - ends[0];
}

// The fragment stage of one chunk.  Its three thread-per-cluster steps each run twice: the lean form (fragment_lean.h) over all clusters, which
// lists the few with a list longer than LEAN_LIST_MAX or a capacity miss, and the general form (aligner.h) over that list, its count read on
// the device (a launch over an empty list is a few microseconds).
static void launchBuildFragments(isaac_gpu_ctx *c, const uint8_t *bcl, u32 clusterBase, u32 n, const isaac_match *matches, const uint64_t *offsets, int withGaps, int trim)
{
    // c->pools: see preparePools
    HIP_CHECK(hipMemsetAsync(c->cigarNext.p, 0, 4, c->stream));
    c->fragTflags.reserve(size_t(GENERAL_BLOCKS) * 16 * 3 * 512); c->fragMatchOrder.reserve(size_t(GENERAL_BLOCKS) * 16 * (MATCH_CAP_MAX + CAND_CAP)); c->indelList.reserve(c->chunkNow);
    c->generalList.reserve(size_t(3) * c->chunkNow); c->generalCount.reserve(4);
    HIP_CHECK(hipMemsetAsync(c->generalCount.p, 0, 16, c->stream));
    u32 *const list0 = c->generalList.p, *const list1 = list0 + c->chunkNow, *const list2 = list1 + c->chunkNow;
    const GappedBuffers gb = gappedBuffers(c, 0);
    HIP_CHECK(hipMemsetAsync(c->gappedCounters.p, 0, 16, c->stream));
    const u32 *order = nullptr;
    AlignList al; al.cap = 8 * c->chunkNow; c->alignList.reserve(al.cap); al.entries = c->alignList.p; al.counter = c->gappedCounters.p + 3;
    {
        ScopedTimer t(c, "build_fragments");
        // (in the order of their match counts this kernel is slower: neighbouring threads no longer read neighbouring matches)
        k_build_fragments<<<gridFor(n, 64), 64, 0, c->stream>>>(c->P, bcl, clusterBase, n, reinterpret_cast<const Match *>(matches), offsets, trim, c->pools, al, list0, c->generalCount.p);
    }
    {
        ScopedTimer t(c, "build_fragments_general");
        k_build_fragments_general<<<GENERAL_BLOCKS, 64, 0, c->stream>>>(c->P, bcl, clusterBase, reinterpret_cast<const Match *>(matches), offsets, trim, c->fragMatchOrder.p, c->fragMatchOrder.p + size_t(GENERAL_BLOCKS) * 16 * MATCH_CAP_MAX, c->pools, al, list0, c->generalCount.p);
        HIP_CHECK(hipGetLastError());
    }
    if (c->P.adapters)
    {   // (--default-adapters only: where each read's adapter lies, before any candidate is aligned)
        ScopedTimer t(c, "adapter_ranges");
        k_adapter_ranges<<<gridFor(u64(4) * n, 256), 256, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, n, c->pools);
        HIP_CHECK(hipGetLastError());
    }
    {
        ScopedTimer t(c, "align_candidates");
        k_align_candidates<<<4096, 256, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, c->pools, al, c->counters.p);
        HIP_CHECK(hipGetLastError());
    }
    {
        ScopedTimer t(c, "finish_candidates");
#if ISAAC_CLUSTER_ORDER
        order = orderClustersByKind(c, n, nullptr);
        c->orderedClusters = n; c->orderedFor = bcl; c->orderedBase = clusterBase;
#endif
        k_finish_candidates<<<gridFor(n, 64), 64, 0, c->stream>>>(c->P, n, withGaps, c->indelList.p, c->gappedCounters.p + 2, c->pools, gb, order, list1, c->generalCount.p + 1);
    }
    {
        ScopedTimer t(c, "finish_candidates_general");
        k_finish_candidates_general<<<GENERAL_BLOCKS, 64, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, withGaps, c->indelList.p, c->gappedCounters.p + 2, c->pools, gb, c->counters.p, list1, c->generalCount.p + 1);
        HIP_CHECK(hipGetLastError());
    }
    {
        ScopedTimer t(c, "indel_fragments");
        k_indel_fragments<<<8192, 64, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, withGaps, c->indelList.p, c->gappedCounters.p + 2, c->pools, gb, c->counters.p);
        HIP_CHECK(hipGetLastError());
    }
    if (withGaps) launchGappedJobs(c, bcl, clusterBase, gb, "gapped_fragments", "gapped_fragments_rescan");
    {
        ScopedTimer t(c, "finish_fragments");
        k_finish_fragments<<<gridFor(n, 64), 64, 0, c->stream>>>(c->P, n, withGaps, c->pools, gb, c->counters.p, order, list2, c->generalCount.p + 2);
    }
    {
        ScopedTimer t(c, "finish_fragments_general");
        k_finish_fragments_general<<<GENERAL_BLOCKS, 64, 0, c->stream>>>(c->P, c->ref(), bcl, clusterBase, withGaps, c->fragTflags.p, c->pools, gb, c->counters.p, list2, c->generalCount.p + 2);
        HIP_CHECK(hipGetLastError());
    }
}

int isaac_gpu_build_fragments(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_match *matches, const uint64_t *offsets,
                              int withGaps, int trim, isaac_candidate *candidates, uint64_t capacity, uint64_t *nCandidatesOut,
                              uint32_t *cigar, uint64_t cigarCapacity, uint64_t *nCigarOut)
{
    ISAAC_TRY
    (void)tile;
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const u32 chunk = chunkFor(c, nClusters);
    preparePools(c, nClusters ? offsetsSpan(c, offsets, nClusters) : 0);
    DevBuf<u32> nc, ng, oc, og; nc.reserve(chunk); ng.reserve(chunk); oc.reserve(chunk); og.reserve(chunk);
    u64 candBase = 0, cigarBase = 0;
    for (u32 done = 0; done < nClusters; done += chunk)
    {
        const u32 n = std::min(chunk, nClusters - done);
        launchBuildFragments(c, bcl, done, n, matches, offsets, withGaps, trim);
        if (!candidates) continue;
        k_count_candidates<<<gridFor(n, 256), 256, 0, st>>>(c->pools, n, nc.p, ng.p);
        exclusiveSum(c, nc.p, oc.p, n); exclusiveSum(c, ng.p, og.p, n);
        k_write_candidates<<<gridFor(n, 256), 256, 0, st>>>(c->pools, done, n, oc.p, og.p, candBase, cigarBase, candidates, capacity, cigar, cigarCapacity);
        u32 a[4];
        HIP_CHECK(hipMemcpyAsync(a + 0, oc.p + n - 1, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(a + 1, nc.p + n - 1, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(a + 2, og.p + n - 1, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(a + 3, ng.p + n - 1, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        candBase += u64(a[0]) + a[1]; cigarBase += u64(a[2]) + a[3];
    }
    HIP_CHECK(hipStreamSynchronize(st));
    if (nCandidatesOut) *nCandidatesOut = candBase;
    if (nCigarOut) *nCigarOut = cigarBase;
    if (candidates && (candBase > capacity || cigarBase > cigarCapacity)) return fail(ISAAC_GPU_ECAPACITY, "candidate buffers are too small");
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_determine_tls(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_match *matches, const uint64_t *offsets, isaac_tls *out)
{
    ISAAC_TRY
    (void)tile;
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    TlsLearner learner(c->P.mateDriftRange);
    if (2 == c->P.nReads)
    {
        const u32 chunk = std::min<u32>(chunkFor(c, std::min<u32>(nClusters, 65536)), 65536);
        c->tlsSamples.reserve(chunk);
        preparePools(c, 0);
        std::vector<TlsSample> h(chunk);
        for (u32 done = 0; done < nClusters && !learner.stats.stable; done += chunk)
        {
            const u32 n = std::min(chunk, nClusters - done);
            launchBuildFragments(c, bcl, done, n, matches, offsets, 0, 0);     // MatchSelector.cpp:233-245: no gaps, no quality trimming
            k_tls_samples<<<gridFor(n, 256), 256, 0, st>>>(c->pools, offsets, done, n, c->tlsSamples.p);
            HIP_CHECK(hipMemcpyAsync(h.data(), c->tlsSamples.p, size_t(n) * sizeof(TlsSample), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            for (u32 i = 0; i < n && !learner.stats.stable; ++i) learner.add(h[i]);
        }
        if (!learner.stats.stable) learner.finalize();
    }
    std::memcpy(out, &learner.stats, sizeof(*out));
    return 0;
    ISAAC_CATCH
}

// what fills the chunk's ClusterFragments: the fragment stage on match lists (isaac_gpu_select) or caller-supplied candidates
struct FragmentSource { const isaac_match *matches; const uint64_t *offsets; const isaac_candidate *candidates; const uint64_t *candidateOffsets; const uint32_t *candidateCigars;
                        uint64_t nMatches; bool nMatchesKnown; };
} // extern "C"
__global__ void k_set_template_constants(TemplateConstants k, TemplateConstants *dst) { *dst = k; }
__global__ void k_widen_sizes(const u32 *sizes, u64 n, u64 *out) { const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) out[i] = sizes[i]; }

// the wave-per-cluster pass over `list` (count on the device) for the chunk described by `p`, on the context's stream
static void launchHeavy(isaac_gpu_ctx *c, const isaac_gpu_ctx::ChunkDesc &p, const u32 *list, const u32 *countDev, u32 blocks, const char *timer, bool sumsKnown)
{
    const TemplateCaps heavy = heavyCaps();
    const u64 heavyBytes = templateWorkBytes(heavy);
    c->heavyArena.reserve(size_t(1024) * heavyBytes);
    RescueBuffers rb; std::memset(&rb, 0, sizeof(rb));
    rb.jobs = c->jobs.p; rb.shadowCands = c->shadowCands.p; rb.candSummaries = c->candSummaries.p; rb.candRank = c->candRank.p; rb.shadowCigars = c->shadowCigars.p; rb.jobBase = c->jobBase.p; rb.jobCount = c->jobCount.p;
    ScopedTimer tm(c, timer);
    k_select_heavy<<<std::max(1u, blocks), 64, HEAVY_SORT_LDS * 2, c->stream>>>(c->P, c->ref(), p.tls, p.rog, logMismatchQ40(), p.bcl, p.clusterBase, 0, countDev, p.tile, c->pools, c->heavyArena.p, heavyBytes, heavy,
                                                                                 list, rb, c->rescueGappedResults.p, c->rescueGappedJobs.p, sumsKnown ? c->clusterSums.p : nullptr, p.records, p.cigars, c->counters.p);
    HIP_CHECK(hipGetLastError());
}

static int selectFromSource(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const FragmentSource &source, const isaac_tls *tls,
                            isaac_fragment *fragments, uint32_t *cigar, uint64_t cigarCapacity)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (cigarCapacity < u64(nClusters) * c->P.nReads * OUT_CIGAR_CAP) return fail(ISAAC_GPU_ECAPACITY, "cigar_dev needs n_clusters * n_reads * ISAAC_GPU_MAX_CIGAR_OPS words");
    hipStream_t st = c->stream;
    DevTls t; std::memcpy(&t, tls, sizeof(t));
    const RogCorrection rog = makeRogCorrection(c->P, c->hContigOffset.data(), c->hContigLoaded.data(), c->nContigs);
    const double lmq40 = logMismatchQ40();
    const u32 chunk = chunkFor(c, nClusters);
    if (chunk > c->selectCapacity)
    {   // the chunk buffers are about to be reallocated: nothing may be left that reads them
        HIP_CHECK(hipStreamSynchronize(st));
        c->selectCapacity = chunk;
    }
    c->overflowList.reserve(chunk);
    RescueBuffers rb; std::memset(&rb, 0, sizeof(rb));
    rb.jobsCap = 4 * chunk; rb.bitmapCap = 64 * rb.jobsCap; rb.candRegionSize = (24 * chunk + CAND_REGIONS - 1) / CAND_REGIONS; rb.candCap = rb.candRegionSize * CAND_REGIONS;
    c->jobs.reserve(rb.jobsCap); c->jobActive.reserve(size_t(rb.jobsCap) + 4); c->longJobs.reserve(rb.jobsCap); c->bitmaps.reserve(rb.bitmapCap); c->candPositions.reserve(rb.candCap); c->candJob.reserve(rb.candCap);
    c->shadowCands.reserve(rb.candCap); c->candSummaries.reserve(rb.candCap); c->candRank.reserve(rb.candCap); c->shadowCigars.reserve(size_t(rb.candCap) * 3); c->jobBase.reserve(chunk); c->jobCount.reserve(chunk); c->rescueCounters.reserve(4 + 2 * CAND_REGIONS);
    rb.jobs = c->jobs.p; rb.jobActive = c->jobActive.p; rb.bitmaps = c->bitmaps.p; rb.candPositions = c->candPositions.p; rb.candJob = c->candJob.p; rb.shadowCands = c->shadowCands.p; rb.candSummaries = c->candSummaries.p; rb.candRank = c->candRank.p;
    rb.shadowCigars = c->shadowCigars.p; rb.jobBase = c->jobBase.p; rb.jobCount = c->jobCount.p;
    rb.jobCounter = c->rescueCounters.p; rb.bitmapCounter = c->rescueCounters.p + 1; rb.candCounter = c->rescueCounters.p + 4;
    c->clusterSums.reserve(chunk); c->heavyList.reserve(chunk); c->mediumList.reserve(chunk); c->largeList.reserve(chunk); c->midList.reserve(chunk); c->hugeList.reserve(chunk); c->xlList.reserve(chunk); c->heavyCount.reserve(16); c->heavyFlag.reserve(chunk);
    c->hugeKeys.reserve(size_t(SUMS_HUGE_BLOCKS) * SUMS_HUGE_CAP * SUMS_HUGE_ENTRY);
    SumsBuffers sb; sb.sums = c->clusterSums.p; sb.residualFlag = c->heavyFlag.p; sb.residualList = c->heavyList.p; sb.residualCount = c->heavyCount.p;
    sb.mediumList = c->mediumList.p; sb.mediumCount = c->heavyCount.p + 4; sb.midList = c->midList.p; sb.midCount = c->heavyCount.p + 8; sb.largeList = c->largeList.p; sb.largeCount = c->heavyCount.p + 1; sb.hugeList = c->hugeList.p; sb.hugeCount = c->heavyCount.p + 2; sb.xlList = c->xlList.p; sb.xlCount = c->heavyCount.p + 3; sb.hugeKeys = c->hugeKeys.p;
    const DevReference R = c->ref();
    {
        TemplateConstants k; k.P = c->P; k.tls = t; k.rog = rog;
        c->templateConstants.reserve(1);
        k_set_template_constants<<<1, 1, 0, st>>>(k, c->templateConstants.p);      // by value: no host buffer to keep alive, no host wait
        HIP_CHECK(hipGetLastError());
    }
    const GappedBuffers gbRescue = gappedBuffers(c, 1);
    // A call whose CIGAR arena runs out (its extra regions: cigarExtra words per cluster) is repeated with four times as many, the work counters put
    // back to where they were: the caller sees one call with exact results.  (Not with deferred completion, where nothing is known yet when the call returns.)
    c->countersSaved.reserve(COUNTER_SHARDS);
    for (int attempt = 0; ; ++attempt)
    {
    if (!c->deferredCompletion) HIP_CHECK(hipMemcpyAsync(c->countersSaved.p, c->counters.p, COUNTER_SHARDS * sizeof(Counters), hipMemcpyDeviceToDevice, st));
    {   // how many candidate slots a chunk can need: the tile's match count -- given by the caller (isaac_gpu_select_n: no host wait), or read from
        // the offsets -- or the caller's candidate count for explicit lists.  A count that is too small shows as ISAAC_GPU_ECAPACITY (poolShort).
        u64 slots = 0;
        if (source.candidates) slots = nClusters ? offsetsSpan(c, source.candidateOffsets, nClusters) : 0;
        else if (source.nMatchesKnown) slots = source.nMatches;
        else slots = nClusters ? offsetsSpan(c, source.offsets, nClusters) : 0;
        preparePools(c, std::max<u64>(slots, 1));
    }
    for (u32 done = 0; done < nClusters; done += chunk)
    {
        const u32 n = std::min(chunk, nClusters - done);
        if (source.candidates)
        {
            gappedBuffers(c, 0);
            HIP_CHECK(hipMemsetAsync(c->gappedCounters.p, 0, 16, st));      // launchBuildFragments does this on the other path
            HIP_CHECK(hipMemsetAsync(c->cigarNext.p, 0, 4, st));
            ScopedTimer tm(c, "load_candidates");
            k_load_candidates<<<gridFor(n, 256), 256, 0, st>>>(c->P, bcl, done, n, source.candidates, source.candidateOffsets, source.candidateCigars, 1, c->pools);
            HIP_CHECK(hipGetLastError());
        }
        else launchBuildFragments(c, bcl, done, n, source.matches, source.offsets, 1, 1);
        HIP_CHECK(hipMemsetAsync(c->overflowCount.p, 0, 4, st));
        HIP_CHECK(hipMemsetAsync(c->rescueCounters.p, 0, (4 + CAND_REGIONS) * 4, st));
        HIP_CHECK(hipMemsetAsync(c->rescueCounters.p + 4 + CAND_REGIONS, 0xff, CAND_REGIONS * 4, st));   // per region: first request that did not fit
        HIP_CHECK(hipMemsetAsync(c->heavyCount.p, 0, 64, st));
        HIP_CHECK(hipMemsetAsync(c->heavyFlag.p, 0, n, st));
        const u32 *order = nullptr;
        {
            ScopedTimer tm(c, "plan_rescue");
#if ISAAC_CLUSTER_ORDER && ISAAC_PLAN_ORDER
            // the order the fragment stage made for this chunk is taken as it is: what the stage did to the lists since (duplicates dropped, gapped alignments) moves
            // few clusters to another kind, and nothing but the waves' uniformity depends on the order (ISAAC_GPU_PLAN_REORDER=1 makes it anew: a measurement)
            static const bool reorder = std::getenv("ISAAC_GPU_PLAN_REORDER") && std::atoi(std::getenv("ISAAC_GPU_PLAN_REORDER"));
            if (!reorder && c->orderedClusters == n && c->orderedFor == bcl && c->orderedBase == done && c->clusterOrder.p) order = c->clusterOrder.p;
            else order = orderClustersByKind(c, n, nullptr);
#endif
            k_plan_rescue<<<gridFor(n, SELECT_BLOCK), SELECT_BLOCK, 0, st>>>(c->templateConstants.p, R, done, n, c->pools, rb, order);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "rescue_windows");
            k_rescue_windows<<<gridFor(rb.jobsCap, RW_WAVES), 64 * RW_WAVES, 0, st>>>(c->P, R, c->hContigOffset[c->nContigs], bcl, done, rb);
            HIP_CHECK(hipGetLastError());
        }
        if (c->P.adapters)
        {   // (--default-adapters only: the shadow strand's adapter of every rescue, from its first candidate position)
            ScopedTimer tm(c, "rescue_adapter_ranges");
            k_rescue_adapter_ranges<<<gridFor(rb.jobsCap, 256), 256, 0, st>>>(c->P, R, bcl, done, rb);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "rescue_align");
            k_rescue_align<<<gridFor(rb.candCap, 256), 256, 0, st>>>(c->P, R, bcl, done, c->pools, rb, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "rescue_gapped_plan");
            k_rescue_gapped_plan<<<gridFor(rb.jobsCap, 256), 256, 0, st>>>(c->pools, rb, gbRescue, c->longJobs.p, c->rescueCounters.p + 2, c->counters.p);
            k_rescue_gapped_plan_long<<<2048, 256, 0, st>>>(c->pools, rb, gbRescue, c->longJobs.p, c->rescueCounters.p + 2);
            HIP_CHECK(hipGetLastError());
        }
        launchGappedJobs(c, bcl, done, gbRescue, "gapped_rescue", "gapped_rescue_rescan");
        {
            ScopedTimer tm(c, "sums_wave");
#if ISAAC_CLUSTER_ORDER
            // from here on the kinds know the clusters' rescue problems as well (k_cluster_sums16, k_select)
            order = orderClustersByKind(c, n, rb.jobCount);
            c->orderedClusters = 0;
#endif
            k_cluster_sums16<<<gridFor(n, SUMS16_GROUPS), 16 * SUMS16_GROUPS, 0, st>>>(c->P, c->pools, n, rb, gbRescue, sb, c->counters.p, order);
            k_cluster_sums<<<8192, 256, 0, st>>>(c->P, c->pools, rb, gbRescue, sb, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "sums_large");
            k_cluster_sums_mid<<<4096, 256, 0, st>>>(c->P, c->pools, rb, gbRescue, sb, c->counters.p);
            k_cluster_sums_large<<<2048, 256, 0, st>>>(c->P, c->pools, rb, gbRescue, sb, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "sums_xl");
            k_cluster_sums_xl<<<256, 1024, SUMS_XL_LDS, st>>>(c->P, c->pools, rb, gbRescue, sb, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "sums_huge");
            k_cluster_sums_huge<<<SUMS_HUGE_BLOCKS, 1024, 0, st>>>(c->P, c->pools, rb, gbRescue, sb, c->counters.p);
            HIP_CHECK(hipGetLastError());
        }
        {
            ScopedTimer tm(c, "select");
            k_select<<<gridFor(n, SELECT_BLOCK), SELECT_BLOCK, 0, st>>>(c->templateConstants.p, R, lmq40, bcl, done, n, tile, c->pools, rb, gbRescue.results, c->clusterSums.p,
                                                     reinterpret_cast<FragmentRecord *>(fragments), cigar, c->overflowList.p, c->overflowCount.p, c->heavyFlag.p, c->counters.p, order);
            HIP_CHECK(hipGetLastError());
        }
        // The wave-per-cluster pass, twice, behind k_select on the same stream: for what the sums stage could not do (near ties, lists beyond
        // its capacities, capacity misses of the flat pass) and for the clusters whose placements overflowed k_select's private lists.  Both
        // read their count on the device and are launched whatever it is -- nearly always zero, a few microseconds -- so that the host never
        // waits for the GPU inside a call: it used to read the two counts back and decide, which cost an idle gap per call.
        isaac_gpu_ctx::ChunkDesc chunkDesc;
        chunkDesc.bcl = bcl; chunkDesc.clusterBase = done; chunkDesc.tile = tile; 
        chunkDesc.records = reinterpret_cast<FragmentRecord *>(fragments); chunkDesc.cigars = cigar; chunkDesc.tls = t; chunkDesc.rog = rog;
        launchHeavy(c, chunkDesc, c->heavyList.p, c->heavyCount.p, 1024u, "select_heavy", false);
        launchHeavy(c, chunkDesc, c->overflowList.p, c->overflowCount.p, 1024u, "select_residual", true);
    }
    if (c->deferredCompletion) return 0;      // the caller enqueues its next call behind this one; isaac_gpu_synchronize waits for the last one
    HIP_CHECK(hipStreamSynchronize(st));
    u32 flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, c->poolShort.p, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    debugTiers(c);
    if (!(flag & 2) || (flag & 1) || attempt >= 3 || c->cigarExtra >= 2048) break;
    c->cigarExtra *= 4;
    HIP_CHECK(hipMemsetAsync(c->poolShort.p, 0, 4, st));
    HIP_CHECK(hipMemcpyAsync(c->counters.p, c->countersSaved.p, COUNTER_SHARDS * sizeof(Counters), hipMemcpyDeviceToDevice, st));
    }
    return checkPoolShort(c);
    ISAAC_CATCH
}
extern "C" {
int isaac_gpu_select(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_match *matches, const uint64_t *offsets, const isaac_tls *tls,
                     isaac_fragment *fragments, uint32_t *cigar, uint64_t cigarCapacity)
{
    FragmentSource source; std::memset(&source, 0, sizeof(source)); source.matches = matches; source.offsets = offsets;
    return selectFromSource(c, bcl, nClusters, tile, source, tls, fragments, cigar, cigarCapacity);
}
int isaac_gpu_select_n(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_match *matches, uint64_t nMatches, const uint64_t *offsets, const isaac_tls *tls,
                       isaac_fragment *fragments, uint32_t *cigar, uint64_t cigarCapacity)
{
    FragmentSource source; std::memset(&source, 0, sizeof(source)); source.matches = matches; source.offsets = offsets; source.nMatches = nMatches; source.nMatchesKnown = true;
    return selectFromSource(c, bcl, nClusters, tile, source, tls, fragments, cigar, cigarCapacity);
}
int isaac_gpu_select_candidates(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_candidate *candidates, const uint64_t *candidateOffsets,
                                const uint32_t *candidateCigars, const isaac_tls *tls, isaac_fragment *fragments, uint32_t *cigar, uint64_t cigarCapacity)
{
    if (!candidates || !candidateOffsets) return fail(ISAAC_GPU_EINVAL, "candidates_dev and candidate_offsets_dev are required");
    FragmentSource source; std::memset(&source, 0, sizeof(source)); source.candidates = candidates; source.candidateOffsets = candidateOffsets; source.candidateCigars = candidateCigars;
    return selectFromSource(c, bcl, nClusters, tile, source, tls, fragments, cigar, cigarCapacity);
}

} // extern "C"
// ---- isaac_gpu_resolve_flagged: the clusters whose MAPQ arithmetic came within 1e-11 of an integer, redone on the host with glibc (resolve_host.cpp)
namespace isaac_host_resolve
{
struct Resolver;
Resolver *create(const isaac_params &params, const char *bases, const u64 *contigOffset, const u8 *contigLoaded, u32 nContigs);
void destroy(Resolver *r);
void selectCluster(Resolver *r, const isaac_tls &tls, const u8 *clusterBcl, u32 cluster, u32 tile, const Match *matches, u32 nMatches, FragmentRecord *records, u32 *cigars);
}
// the clusters of a tile with RECORD_MAPQ_NEAR_INTEGER on their first record, in any order (the list has room for every cluster of the tile)
__global__ void k_flagged_clusters(const FragmentRecord *records, u32 nClusters, u32 nReads, u32 *list, u32 *count)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nClusters || !(records[u64(c) * nReads].reserved & RECORD_MAPQ_NEAR_INTEGER)) return;
    list[atomicAdd(count, 1u)] = c;
}
// What the host wants of the listed clusters, gathered into one staging buffer so that it leaves in one copy: per cluster `stride` bytes --
// the cluster's match range (16), its BCL bytes (padded to 16), its records, its CIGAR slots.  One workgroup per cluster.
__global__ void k_gather_flagged(const u32 *list, u32 n, const u8 *bcl, u32 clusterLength, const u64 *offsets, const FragmentRecord *records, const u32 *cigar, u32 nReads, u8 *staging, u32 stride)
{
    const u32 k = blockIdx.x;
    if (k >= n) return;
    const u32 c = list[k];
    u8 *to = staging + u64(k) * stride;
    if (threadIdx.x < 2) reinterpret_cast<u64 *>(to)[threadIdx.x] = offsets[c + threadIdx.x];
    to += 16;
    for (u32 i = threadIdx.x; i < clusterLength; i += blockDim.x) to[i] = bcl[u64(c) * clusterLength + i];
    to += (clusterLength + 15) & ~15u;
    const u32 recordWords = nReads * u32(sizeof(FragmentRecord) / 4), cigarWords = nReads * OUT_CIGAR_CAP;
    const u32 *r = reinterpret_cast<const u32 *>(records + u64(c) * nReads), *g = cigar + u64(c) * cigarWords;
    for (u32 i = threadIdx.x; i < recordWords; i += blockDim.x) reinterpret_cast<u32 *>(to)[i] = r[i];
    to += recordWords * 4;
    for (u32 i = threadIdx.x; i < cigarWords; i += blockDim.x) reinterpret_cast<u32 *>(to)[i] = g[i];
}
// the listed clusters' seed matches back to back: ranges[2k], ranges[2k + 1] = where cluster k's lie in the tile's array, starts[k] = where they go
__global__ void k_gather_flagged_matches(const u64 *ranges, const u64 *starts, u32 n, const Match *matches, Match *out)
{
    const u32 k = blockIdx.x;
    if (k >= n) return;
    const u64 first = ranges[2 * k], count = ranges[2 * k + 1] - first;
    for (u64 i = threadIdx.x; i < count; i += blockDim.x) out[starts[k] + i] = matches[first + i];
}
// ... and the way back for the clusters the host changed: records and CIGAR slots from `staging` (stride bytes per cluster: records, then CIGAR slots)
__global__ void k_scatter_resolved(const u32 *list, u32 n, const u8 *staging, u32 stride, FragmentRecord *records, u32 *cigar, u32 nReads)
{
    const u32 k = blockIdx.x;
    if (k >= n) return;
    const u32 c = list[k];
    const u32 recordWords = nReads * u32(sizeof(FragmentRecord) / 4), cigarWords = nReads * OUT_CIGAR_CAP;
    const u32 *from = reinterpret_cast<const u32 *>(staging + u64(k) * stride);
    u32 *r = reinterpret_cast<u32 *>(records + u64(c) * nReads), *g = cigar + u64(c) * cigarWords;
    for (u32 i = threadIdx.x; i < recordWords; i += blockDim.x) r[i] = from[i];
    for (u32 i = threadIdx.x; i < cigarWords; i += blockDim.x) g[i] = from[recordWords + i];
}
// the caller's own copy of the contigs (as given to isaac_gpu_load_contigs), kept by the caller for as long as the context lives: isaac_gpu_resolve_flagged then
// has nothing to fetch
extern "C" int isaac_gpu_set_host_contigs(isaac_gpu_ctx *c, const char *basesHost)
{
    ISAAC_TRY
    dropResolvers(c, true);
    c->hostBasesGiven = basesHost;
    return 0;
    ISAAC_CATCH
}
extern "C" int isaac_gpu_resolve_flagged(isaac_gpu_ctx *c, const uint8_t *bcl, uint32_t nClusters, uint32_t tile, const isaac_match *matches, const uint64_t *offsets, const isaac_tls *tls,
                                         isaac_fragment *fragments, uint32_t *cigar, uint64_t *nFlaggedOut, uint64_t *nChangedOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nFlaggedOut) *nFlaggedOut = 0;
    if (nChangedOut) *nChangedOut = 0;
    if (!nClusters) return 0;
    if (!bcl || !matches || !offsets || !tls || !fragments || !cigar) return fail(ISAAC_GPU_EINVAL, "bcl_dev, matches_dev, cluster_offsets_dev, tls, fragments_dev and cigar_dev are required");
    hipStream_t st = c->stream;
    const u32 nReads = c->P.nReads, clusterLength = c->P.clusterLength;
    c->flaggedList.reserve(size_t(nClusters) + 1);
    u32 *listCount = c->flaggedList.p + nClusters;
    HIP_CHECK(hipMemsetAsync(listCount, 0, 4, st));
    FragmentRecord *records = reinterpret_cast<FragmentRecord *>(fragments);
    k_flagged_clusters<<<gridFor(nClusters, 256), 256, 0, st>>>(records, nClusters, nReads, c->flaggedList.p, listCount);
    HIP_CHECK(hipGetLastError());
    u32 nAll = 0;
    HIP_CHECK(hipMemcpyAsync(&nAll, listCount, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (nFlaggedOut) *nFlaggedOut = nAll;
    if (!nAll) return 0;
    std::vector<u32> list(nAll);
    HIP_CHECK(hipMemcpy(list.data(), c->flaggedList.p, size_t(nAll) * 4, hipMemcpyDeviceToHost));
    std::sort(list.begin(), list.end());
    HIP_CHECK(hipMemcpy(c->flaggedList.p, list.data(), size_t(nAll) * 4, hipMemcpyHostToDevice));
    // the host's copy of the contigs: the caller's (isaac_gpu_set_host_contigs), or fetched once, when the first flagged cluster turns up
    const char *hostBases = c->hostBasesGiven;
    if (!hostBases)
    {
        if (c->hostBases.empty() && c->hContigOffset[c->nContigs])
        {
            c->hostBases.resize(c->hContigOffset[c->nContigs]);
            HIP_CHECK(hipMemcpy(c->hostBases.data(), c->bases, c->hostBases.size(), hipMemcpyDeviceToHost));
        }
        hostBases = c->hostBases.data();
    }
    if (c->resolverLoaded != c->hContigLoaded)
    {   // (the rest-of-genome correction depends on which contigs count as loaded)
        dropResolvers(c, false);
        c->resolverLoaded = c->hContigLoaded;
    }
    // In batches (a tile rich in ties -- equal placements are what makes probability ratios exact powers of ten -- may flag many): everything the batch's clusters
    // need is gathered on the device and fetched in two copies; then the clusters side by side on host threads (a cluster of a repeat family is milliseconds of
    // serial mate rescue); then what changed goes back in one copy and one kernel.
    const u32 batchMax = 1u << 16;
    const u32 bclBytes = (clusterLength + 15) & ~15u, recordBytes = nReads * u32(sizeof(FragmentRecord)), cigarBytes = nReads * OUT_CIGAR_CAP * 4;
    const u32 stride = 16 + bclBytes + recordBytes + cigarBytes, backStride = recordBytes + cigarBytes;
    DevBuf<u8> &staging = c->resolveStaging, &back = c->resolveBack; DevBuf<u64> &ranges = c->resolveRanges; DevBuf<Match> &gathered = c->resolveMatches; DevBuf<u32> &changedList = c->resolveChanged;
    const u32 nThreads = std::min<u32>(nAll, std::min<u32>(16, std::max(1u, std::thread::hardware_concurrency())));
    while (c->resolvers.size() < nThreads) c->resolvers.push_back(isaac_host_resolve::create(c->params, hostBases, c->hContigOffset.data(), c->resolverLoaded.data(), c->nContigs));
    u64 changed = 0;
    for (u32 done = 0; done < nAll; done += batchMax)
    {
        const u32 n = std::min(batchMax, nAll - done);
        const u32 *batchList = c->flaggedList.p + done;
        staging.reserve(size_t(n) * stride);
        k_gather_flagged<<<n, 64, 0, st>>>(batchList, n, bcl, clusterLength, offsets, records, cigar, nReads, staging.p, stride);
        HIP_CHECK(hipGetLastError());
        std::vector<u8> host(size_t(n) * stride);
        HIP_CHECK(hipMemcpyAsync(host.data(), staging.p, host.size(), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<u64> hRanges(2 * size_t(n)), starts(size_t(n) + 1, 0);
        for (u32 k = 0; k < n; ++k)
        {
            std::memcpy(&hRanges[2 * k], host.data() + size_t(k) * stride, 16);
            starts[k + 1] = starts[k] + (hRanges[2 * k + 1] - hRanges[2 * k]);
        }
        std::vector<Match> hMatches(starts[n] + 1);
        if (starts[n])
        {
            ranges.reserve(3 * size_t(n)); gathered.reserve(starts[n]);
            HIP_CHECK(hipMemcpyAsync(ranges.p, hRanges.data(), 2 * size_t(n) * 8, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(ranges.p + 2 * size_t(n), starts.data(), size_t(n) * 8, hipMemcpyHostToDevice, st));
            k_gather_flagged_matches<<<n, 64, 0, st>>>(ranges.p, ranges.p + 2 * size_t(n), n, reinterpret_cast<const Match *>(matches), gathered.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(hMatches.data(), gathered.p, starts[n] * sizeof(Match), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        std::vector<u8> now(size_t(n) * backStride, 0);
        std::vector<u8> differs(n, 0); std::vector<std::string> errors(nThreads);
        std::atomic<u32> next(0);
        const auto work = [&](u32 t)
        {
            for (u32 k = next++; k < n; k = next++)
            {
                const u32 cluster = list[done + k];
                const u8 *item = host.data() + size_t(k) * stride;
                const u8 *itemBcl = item + 16;
                const FragmentRecord *was = reinterpret_cast<const FragmentRecord *>(itemBcl + bclBytes);
                const u32 *wasCigar = reinterpret_cast<const u32 *>(itemBcl + bclBytes + recordBytes);
                FragmentRecord *nowRecords = reinterpret_cast<FragmentRecord *>(now.data() + size_t(k) * backStride);
                u32 *nowCigar = reinterpret_cast<u32 *>(now.data() + size_t(k) * backStride + recordBytes);
                try
                {
                    isaac_host_resolve::selectCluster(c->resolvers[t], *tls, itemBcl, cluster, tile, hMatches.data() + starts[k], u32(starts[k + 1] - starts[k]), nowRecords, nowCigar);
                    for (u32 r = 0; r < nReads; ++r)
                    {
                        FragmentRecord a = was[r], b = nowRecords[r];
                        // the diagnostic bits apart (the host form does not flag): everything the record says, and its CIGAR
                        b.reserved = (b.reserved & 0xffff0000u) | (a.reserved & 0xffffu & ~u32(RECORD_NOT_STORED | RECORD_FRAGMENT_OVERFLOW)) | (b.reserved & (RECORD_NOT_STORED | RECORD_FRAGMENT_OVERFLOW));
                        nowRecords[r] = b;
                        if (std::memcmp(&a, &b, sizeof(a))) differs[k] = 1;
                        const u64 base = u64(cluster) * nReads * OUT_CIGAR_CAP;
                        const u32 *ca = wasCigar + (a.cigarOffset - base), *cb = nowCigar + (b.cigarOffset - base);
                        if (a.cigarLength == b.cigarLength && std::memcmp(ca, cb, size_t(a.cigarLength) * 4)) differs[k] = 1;
                    }
                }
                catch (const std::exception &e) { errors[t] = e.what(); }
            }
        };
        {
            std::vector<std::thread> threads;
            for (u32 t = 1; t < nThreads; ++t) threads.emplace_back(work, t);
            work(0);
            for (std::thread &t : threads) t.join();
        }
        for (const std::string &e : errors) if (!e.empty()) return fail(ISAAC_GPU_EHIP, "isaac_gpu_resolve_flagged: " + e);
        std::vector<u32> changedClusters; std::vector<u8> changedData;
        for (u32 k = 0; k < n; ++k)
            if (differs[k])
            {
                changedClusters.push_back(list[done + k]);
                changedData.insert(changedData.end(), now.begin() + size_t(k) * backStride, now.begin() + size_t(k + 1) * backStride);
            }
        if (!changedClusters.empty())
        {
            const u32 m = u32(changedClusters.size());
            changedList.reserve(m); back.reserve(changedData.size());
            HIP_CHECK(hipMemcpyAsync(changedList.p, changedClusters.data(), size_t(m) * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(back.p, changedData.data(), changedData.size(), hipMemcpyHostToDevice, st));
            k_scatter_resolved<<<m, 64, 0, st>>>(changedList.p, m, back.p, backStride, records, cigar, nReads);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(st));
            changed += m;
        }
    }
    if (nChangedOut) *nChangedOut = changed;
    return 0;
    ISAAC_CATCH
}

// the CIGARs of a tile's records packed back to back (isaac_gpu_compact_cigars)
__global__ void k_cigar_lengths(const FragmentRecord *records, u64 n, u32 *lengths)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) lengths[i] = records[i].cigarLength;
}
// nothing is touched when the packed pool would not fit: the records keep their slot offsets, so that the call can be repeated
// with a larger pool (the total is offsets[n - 1] + lengths[n - 1], known on the device before this kernel starts)
__global__ void k_cigar_pack(FragmentRecord *records, u64 n, const u32 *offsets, const u32 *lengths, const u32 *cigarIn, u32 *cigarOut, u64 capacity, u64 *totalOut)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    const u64 total = u64(offsets[n - 1]) + lengths[n - 1];
    if (0 == i && totalOut) *totalOut = total;
    if (i >= n || total > capacity) return;
    const u32 from = records[i].cigarOffset, to = offsets[i], len = lengths[i];
    for (u32 k = 0; k < len; ++k) cigarOut[to + k] = cigarIn[from + k];
    records[i].cigarOffset = to;
}
static int compactCigars(isaac_gpu_ctx *c, isaac_fragment *fragments, uint64_t nRecords, const uint32_t *cigarIn, uint32_t *cigarOut, uint64_t capacity, u64 *totalDev)
{
    if (!fragments || !cigarIn || !cigarOut) return fail(ISAAC_GPU_EINVAL, "fragments_dev, cigar_in_dev and cigar_out_dev are required");
    if (nRecords >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "at most 2^31 - 1 records per call");
    hipStream_t st = c->stream;
    DevBuf<u32> &len = c->cigarLengths, &off = c->cigarOffsets; len.reserve(nRecords); off.reserve(nRecords);
    FragmentRecord *records = reinterpret_cast<FragmentRecord *>(fragments);
    k_cigar_lengths<<<gridFor(nRecords, 256), 256, 0, st>>>(records, nRecords, len.p);
    exclusiveSum(c, len.p, off.p, nRecords);
    k_cigar_pack<<<gridFor(nRecords, 256), 256, 0, st>>>(records, nRecords, off.p, len.p, cigarIn, cigarOut, capacity, totalDev);
    HIP_CHECK(hipGetLastError());
    return 0;
}
extern "C" {
int isaac_gpu_compact_cigars(isaac_gpu_ctx *c, isaac_fragment *fragments, uint64_t nRecords, const uint32_t *cigarIn, uint32_t *cigarOut, uint64_t capacity, uint64_t *nWordsOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nWordsOut) *nWordsOut = 0;
    if (!nRecords) return 0;
    c->cigarTotal.reserve(1);
    if (const int rc = compactCigars(c, fragments, nRecords, cigarIn, cigarOut, capacity, c->cigarTotal.p)) return rc;
    u64 total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, c->cigarTotal.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (nWordsOut) *nWordsOut = total;
    if (total > capacity) return fail(ISAAC_GPU_ECAPACITY, "cigar_out_dev is too small");
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_compact_cigars_async(isaac_gpu_ctx *c, isaac_fragment *fragments, uint64_t nRecords, const uint32_t *cigarIn, uint32_t *cigarOut, uint64_t capacity, uint64_t *nWordsOutDev)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (!nWordsOutDev) return fail(ISAAC_GPU_EINVAL, "n_words_out_dev is required");
    if (!nRecords) { HIP_CHECK(hipMemsetAsync(nWordsOutDev, 0, 8, c->stream)); return 0; }
    return compactCigars(c, fragments, nRecords, cigarIn, cigarOut, capacity, reinterpret_cast<u64 *>(nWordsOutDev));
    ISAAC_CATCH
}

// The BAM alignment records of a set of tiles in file order (bam_kernels.h)
int isaac_gpu_bam_records(isaac_gpu_ctx *c, const isaac_bam_tile *tiles, uint32_t nTiles, const isaac_bam_options *options, uint8_t *bam, uint64_t capacity,
                          uint64_t *nBytesOut, uint64_t *nRecordsOut, uint64_t *unalignedOffsetOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nBytesOut) *nBytesOut = 0;
    if (nRecordsOut) *nRecordsOut = 0;
    if (unalignedOffsetOut) *unalignedOffsetOut = 0;
    if (nTiles && !tiles) return fail(ISAAC_GPU_EINVAL, "tiles is required");
    BamOptions o; std::memset(&o, 0, sizeof(o));
    o.nReads = c->params.n_reads;
    for (u32 r = 0; r < o.nReads; ++r) { o.readLength[r] = c->params.read_length[r]; o.readOffset[r] = o.clusterLength; o.clusterLength += o.readLength[r]; }
    const char *readGroup = options && options->read_group ? options->read_group : "0", *barcode = options && options->barcode ? options->barcode : "none";
    if (std::strlen(readGroup) >= sizeof(o.readGroup) || std::strlen(barcode) >= sizeof(o.barcode)) return fail(ISAAC_GPU_EINVAL, "read_group and barcode: at most 63 characters");
    std::strcpy(o.readGroup, readGroup); o.readGroupLength = u32(std::strlen(readGroup)); std::strcpy(o.barcode, barcode); o.barcodeLength = u32(std::strlen(barcode));
    o.forcedDodgyAlignmentScore = options ? (options->forced_dodgy_alignment_score & 0xff) : (u32(c->params.dodgy_alignment_score) & 0xff);
    o.pessimisticMapQ = options ? options->pessimistic_mapq : 0;
    o.markDuplicates = options ? (options->mark_duplicates != 0) : 0; o.keepDuplicates = options ? (options->keep_duplicates != 0) : 1;
    o.realignGaps = options ? (options->realign_gaps != 0) : 0;
    o.indexEntries = options ? options->index_entries_dev : nullptr;
    if (options && options->bin_filter)
    {
        o.binFilter = 2 == options->bin_filter ? 2 : 1; o.binFirstContig = options->bin_first_contig; o.binEndContig = options->bin_end_contig; o.binUnaligned = options->bin_unaligned != 0;
        o.binFirstPosition = options->bin_first_position & ~u64(1); o.binEndPosition = options->bin_end_position & ~u64(1);
        if (2 == o.binFilter && o.binEndPosition < o.binFirstPosition) return fail(ISAAC_GPU_EINVAL, "bin_filter 2: bin_first_position <= bin_end_position (ReferencePosition values)");
    }
    if (o.realignGaps)
    {
        // BinSorter.hh:96-98: GapRealigner(realignGapsVigorously, realignDodgyFragments, realignedGapsPerFragment, 3, 4, 0, clipSemialigned, ...)
        o.realign.realignGapsVigorously = options->realign_vigorously != 0;
        o.realign.mismatchCost = 3; o.realign.gapOpenCost = 4; o.realign.gapExtendCost = 0; o.realign.realignDodgyFragments = options->realign_dodgy != 0; o.realign.clipSemialigned = c->params.clip_semialigned != 0;
        if (options->tls) std::memcpy(&o.tls, options->tls, sizeof(o.tls));
    }
    std::vector<BamTile> h(nTiles);
    u64 n = 0; u32 maxReadGroup = 0;
    for (u32 t = 0; t < nTiles; ++t)
    {
        const isaac_bam_tile &in = tiles[t];
        // (cigar_dev may be NULL for a tile without a single CIGAR word: the unaligned bin's tiles)
        if (in.n_records && (!in.bcl_dev || !in.fragments_dev)) return fail(ISAAC_GPU_EINVAL, "bcl_dev and fragments_dev are required for every tile");
        const char *prefix = in.read_name_prefix ? in.read_name_prefix : "";
        if (std::strlen(prefix) >= sizeof(h[t].name)) return fail(ISAAC_GPU_EINVAL, "read_name_prefix: at most 63 characters");
        std::memset(&h[t], 0, sizeof(BamTile));
        h[t].bcl = in.bcl_dev; h[t].records = reinterpret_cast<const FragmentRecord *>(in.fragments_dev); h[t].cigars = in.cigar_dev; h[t].firstRecord = n;
        h[t].nRecords = u32(in.n_records); h[t].nameLength = u32(std::strlen(prefix)); std::memcpy(h[t].name, prefix, h[t].nameLength);
        const char *tileReadGroup = in.read_group ? in.read_group : readGroup;
        if (std::strlen(tileReadGroup) >= sizeof(h[t].readGroup)) return fail(ISAAC_GPU_EINVAL, "read_group: at most 27 characters");
        h[t].readGroupLength = u32(std::strlen(tileReadGroup)); std::memcpy(h[t].readGroup, tileReadGroup, h[t].readGroupLength);
        maxReadGroup = std::max(maxReadGroup, h[t].readGroupLength);
        if (o.realignGaps && 2 == o.nReads && !in.tls && !options->tls)
            return fail(ISAAC_GPU_EINVAL, "gap realignment of paired reads needs the template length statistics (isaac_bam_options::tls or isaac_bam_tile::tls)");
        if (in.tls) std::memcpy(&h[t].tls, in.tls, sizeof(h[t].tls)); else h[t].tls = o.tls;
        n += in.n_records;
    }
    if (n >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "at most 2^31 - 1 records per call");
    if (!n) return 0;
    if (!bam && capacity) return fail(ISAAC_GPU_EINVAL, "bam_dev is required");
    hipStream_t st = c->stream;
    c->bamTiles.reserve(nTiles); c->bamKeyHi.reserve(n); c->bamKeyLo.reserve(n); c->bamKeyAlt.reserve(n); c->bamOffsets.reserve(n); c->bamBytes64.reserve(n);
    c->bamIndex.reserve(n); c->bamIndexAlt.reserve(n); c->bamBytes.reserve(n); c->bamBounds.reserve(2);
    if (o.realignGaps)
    {   // the realigner works on a copy of the records: positions, CIGARs, TLEN and proper-pair flags change, the caller's buffers do not
        c->realignRecords.reserve(n); c->realignPool.reserve(n + 1024); c->realignNext.reserve(1); c->realignChanged.reserve(n);
        for (u32 t = 0; t < nTiles; ++t)
        {
            if (h[t].nRecords) HIP_CHECK(hipMemcpyAsync(c->realignRecords.p + h[t].firstRecord, h[t].records, sizeof(FragmentRecord) * h[t].nRecords, hipMemcpyDeviceToDevice, st));
            h[t].recordsOriginal = h[t].records; h[t].records = c->realignRecords.p + h[t].firstRecord; h[t].cigarsAlt = c->realignPool.p;
        }
    }
    HIP_CHECK(hipMemcpyAsync(c->bamTiles.p, h.data(), sizeof(BamTile) * nTiles, hipMemcpyHostToDevice, st));
    const u64 bounds0[2] = { n, n };
    HIP_CHECK(hipMemcpyAsync(c->bamBounds.p, bounds0, sizeof(bounds0), hipMemcpyHostToDevice, st));
    const u8 *duplicate = nullptr;
    if (o.markDuplicates || !o.keepDuplicates)
    {   // BinSorter::resolveDuplicates: the verdict per record, before the records are ordered (bam_kernels.h)
        ScopedTimer t(c, "bam_duplicates");
        c->dupPrimary.reserve(n); c->dupMate.reserve(n); c->dupRank.reserve(n); c->dupCluster.reserve(n); c->dupSmall.reserve(n); c->dupFlag.reserve(n);
        k_dup_keys<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, c->dupPrimary.p, c->dupMate.p, c->dupRank.p, c->dupCluster.p, c->dupSmall.p, c->bamIndex.p);
        // stable passes from the least significant key up; the keys of a pass are gathered in the order the previous one left
        u32 *from = c->bamIndex.p, *to = c->bamIndexAlt.p;
        const struct { const u64 *key; int bits; } passes[5] = { { c->dupCluster.p, 64 }, { c->dupRank.p, 64 }, { c->dupMate.p, 64 }, { c->dupPrimary.p, 64 }, { c->dupSmall.p, 4 } };
        for (const auto &pass : passes)
        {
            k_dup_gather<<<gridFor(n, 256), 256, 0, st>>>(pass.key, from, n, c->bamKeyLo.p);
            sortPairs(c, c->bamKeyLo.p, c->bamKeyAlt.p, from, to, n, pass.bits);
            std::swap(from, to);
        }
        HIP_CHECK(hipMemsetAsync(c->dupFlag.p, 0, n, st));
        k_dup_mark<<<gridFor(n, 256), 256, 0, st>>>(from, n, c->dupPrimary.p, c->dupMate.p, c->dupCluster.p, c->dupSmall.p, c->dupFlag.p);
        HIP_CHECK(hipGetLastError());
        duplicate = c->dupFlag.p;
    }
    if (o.realignGaps)
    {   // BinSorter::collectGaps + realignGaps with every contig as one bin (bam_kernels.h, realign.h)
        ScopedTimer t(c, "bam_realign");
        u32 *counts = c->bamBytes.p, *offsets = c->bamIndexAlt.p;
        k_realign_count<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, counts);
        exclusiveSum(c, counts, offsets, n);
        u32 lastCount = 0, lastOffset = 0;
        HIP_CHECK(hipMemcpyAsync(&lastCount, counts + n - 1, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(&lastOffset, offsets + n - 1, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const u64 nGaps = u64(lastCount) + lastOffset;
        std::vector<RealignGap> gaps(nGaps), deletionEnds;
        if (nGaps)
        {
            c->realignGaps.reserve(nGaps); c->realignDeletionEnds.reserve(nGaps);
            k_realign_collect<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, offsets, c->realignGaps.p);
            HIP_CHECK(hipMemcpyAsync(gaps.data(), c->realignGaps.p, nGaps * sizeof(RealignGap), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            // RealignerGaps::finalizeGaps (GapRealigner.cpp:86-94) with the host's std::sort, one contig (bin) at a time for the deletion ends
            std::sort(gaps.begin(), gaps.end(), [](const RealignGap &l, const RealignGap &r) { return rgLess(l, r); });
            gaps.erase(std::unique(gaps.begin(), gaps.end(), [](const RealignGap &l, const RealignGap &r) { return l.pos == r.pos && l.length == r.length; }), gaps.end());
            for (size_t b = 0; b < gaps.size();)
            {
                size_t e = b; while (e < gaps.size() && refposContig(gaps[e].pos) == refposContig(gaps[b].pos)) ++e;
                const size_t first = deletionEnds.size();
                for (size_t k = b; k < e; ++k) if (rgIsDeletion(gaps[k])) deletionEnds.push_back(gaps[k]);
                std::sort(deletionEnds.begin() + first, deletionEnds.end(), [](const RealignGap &l, const RealignGap &r) { return rgEndPos(l, false) < rgEndPos(r, false); });
                b = e;
            }
            HIP_CHECK(hipMemcpyAsync(c->realignGaps.p, gaps.data(), gaps.size() * sizeof(RealignGap), hipMemcpyHostToDevice, st));
            if (!deletionEnds.empty()) HIP_CHECK(hipMemcpyAsync(c->realignDeletionEnds.p, deletionEnds.data(), deletionEnds.size() * sizeof(RealignGap), hipMemcpyHostToDevice, st));
        }
        RealignerGapsView view = { c->realignGaps.p, u32(gaps.size()), c->realignDeletionEnds.p, u32(deletionEnds.size()) };
        // The new CIGARs take words from a pool by a bump counter that keeps counting when the pool is full: a pass that did not fit (deep or
        // indel-rich bins: more than a word per record on average) is repeated on a fresh copy of the records with a pool of the size it asked for,
        // so that no realignment is ever dropped and the outcome does not depend on which fragments came first.
        u64 poolCap = c->realignPool.n;
        if (const char *small = std::getenv("ISAAC_GPU_REALIGN_POOL_WORDS")) poolCap = std::min<u64>(poolCap, u64(std::atol(small)));     // tests: the second pass
        // the fragments with a gap of the list in their range (k_realign_filter), then the realigner over those
        c->realignList.reserve(n);
        k_realign_filter<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, view, duplicate, c->realignRecords.p, counts, c->realignChanged.p);       // (counts, offsets: free again after the gaps' collection)
        HIP_CHECK(hipGetLastError());
        exclusiveSum(c, counts, offsets, n);
        k_realign_list<<<gridFor(n, 256), 256, 0, st>>>(counts, offsets, n, c->realignList.p);
        HIP_CHECK(hipGetLastError());
        u32 listCount = 0;
        {
            u32 lastWanted = 0, lastAt = 0;
            HIP_CHECK(hipMemcpyAsync(&lastWanted, counts + n - 1, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipMemcpyAsync(&lastAt, offsets + n - 1, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));         // (the host vectors above are read by the copies)
            listCount = lastWanted + lastAt;
        }
        for (int attempt = 0; listCount; ++attempt)
        {
            HIP_CHECK(hipMemsetAsync(c->realignNext.p, 0, 4, st));
            k_realign<<<gridFor(listCount, 128), 128, 0, st>>>(c->bamTiles.p, nTiles, c->realignList.p, listCount, o, c->ref(), view, duplicate, c->realignRecords.p, c->realignPool.p, u32(std::min<u64>(poolCap, 0xffffffffu)), c->realignNext.p, c->realignChanged.p);
            HIP_CHECK(hipGetLastError());
            u32 wanted = 0;
            HIP_CHECK(hipMemcpyAsync(&wanted, c->realignNext.p, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));         // (the host vectors above are read by the copies)
            if (wanted <= poolCap) break;
            if (attempt) return fail(ISAAC_GPU_ECAPACITY, "the gap realigner's CIGAR pool overflowed twice");
            c->realignPool.reserve(size_t(wanted) + 1024); poolCap = c->realignPool.n;
            for (u32 t = 0; t < nTiles; ++t)
            {
                if (h[t].nRecords) HIP_CHECK(hipMemcpyAsync(c->realignRecords.p + h[t].firstRecord, h[t].recordsOriginal, sizeof(FragmentRecord) * h[t].nRecords, hipMemcpyDeviceToDevice, st));
                h[t].cigarsAlt = c->realignPool.p;
            }
            HIP_CHECK(hipMemcpyAsync(c->bamTiles.p, h.data(), sizeof(BamTile) * nTiles, hipMemcpyHostToDevice, st));
        }
        if (listCount) k_realign_pairs<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, c->realignRecords.p, c->realignChanged.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    {
        ScopedTimer t(c, "bam_order");
        k_bam_keys<<<gridFor(n, 256), 256, 0, st>>>(c->bamTiles.p, nTiles, n, o, duplicate, c->bamKeyHi.p, c->bamKeyLo.p, c->bamIndex.p, c->bamBytes.p);
        // two stable passes: by (global cluster id, unmapped, second read), then by bin position
        sortPairs(c, c->bamKeyLo.p, c->bamKeyAlt.p, c->bamIndex.p, c->bamIndexAlt.p, n);
        k_bam_gather_hi<<<gridFor(n, 256), 256, 0, st>>>(c->bamKeyHi.p, c->bamIndexAlt.p, n, c->bamKeyLo.p);
        sortPairs(c, c->bamKeyLo.p, c->bamKeyAlt.p, c->bamIndexAlt.p, c->bamIndex.p, n);
        k_bam_gather_bytes<<<gridFor(n, 256), 256, 0, st>>>(c->bamBytes.p, c->bamIndex.p, n, c->bamBytes64.p);
        exclusiveSum(c, c->bamBytes64.p, c->bamOffsets.p, n);
        k_bam_bounds<<<gridFor(n, 256), 256, 0, st>>>(c->bamKeyAlt.p, n, c->bamBounds.p);
    }
    {
        ScopedTimer t(c, "bam_encode");
        // dynamic LDS: the image of a chunk of the file + the staged bases of its records
        u32 maxRead = 0, maxName = 0;
        for (u32 r = 0; r < o.nReads; ++r) maxRead = std::max(maxRead, o.readLength[r]);
        for (u32 t = 0; t < nTiles; ++t) maxName = std::max(maxName, h[t].nameLength);
        BamChunkLds lds;
        lds.segments = (maxRead + 15) / 16;
        lds.chunkBytes = std::min<u32>(bamChunkImageBytes(maxRead, maxName + 13, 28 + maxReadGroup + o.barcodeLength), 96 * 1024);
        const size_t dynamicBytes = (lds.chunkBytes + 16 + 15) & ~15u;
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bam_encode), hipFuncAttributeMaxDynamicSharedMemorySize, int(dynamicBytes)));
        k_bam_encode<<<gridFor(n, BAM_CHUNK_RECORDS), 256, dynamicBytes, st>>>(c->bamTiles.p, nTiles, n, o, c->bamIndex.p, c->bamOffsets.p, c->bamBytes64.p, duplicate, bam, capacity, lds);
    }
    HIP_CHECK(hipGetLastError());
    u64 lastOffset = 0, lastBytes = 0, bounds[2] = { 0, 0 };
    HIP_CHECK(hipMemcpyAsync(&lastOffset, c->bamOffsets.p + n - 1, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&lastBytes, c->bamBytes64.p + n - 1, 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(bounds, c->bamBounds.p, 16, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const u64 total = lastOffset + lastBytes;
    u64 unalignedOffset = total;
    if (bounds[0] < n) HIP_CHECK(hipMemcpy(&unalignedOffset, c->bamOffsets.p + bounds[0], 8, hipMemcpyDeviceToHost));
    if (nBytesOut) *nBytesOut = total;
    if (nRecordsOut) *nRecordsOut = bounds[1];
    if (unalignedOffsetOut) *unalignedOffsetOut = unalignedOffset;
    if (total > capacity) return fail(ISAAC_GPU_ECAPACITY, "bam_dev is too small");
    return 0;
    ISAAC_CATCH
}

// BGZF without compression on the device (bgzf_kernels.h)
// ---- isaac_gpu_bin_tile (bam_kernels.h: k_bin_*)
int isaac_gpu_bin_tile_map(isaac_gpu_ctx *c, const uint8_t *bcl, const isaac_fragment *fragments, const uint32_t *cigars, uint32_t nClusters, const isaac_bin_map *map,
                           uint8_t *out, uint64_t capacity, isaac_bin_size *sizesOut, uint64_t *nBytesOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nBytesOut) *nBytesOut = 0;
    if (!map || !map->n_bins || map->n_bins > BIN_MAX || !sizesOut || (map->n_contigs && !map->bin_of_contig) || (map->n_cuts && !map->cut_positions))
        return fail(ISAAC_GPU_EINVAL, "1 .. 65535 bins, sizes_out, bin_of_contig and cut_positions are required");
    const u32 nBins = map->n_bins, nContigs = map->n_contigs, nCuts = map->n_cuts;
    {   // every contig's bins end below the last bin (which takes the templates without a position); the cuts ascend
        std::vector<u32> cutsOf(nContigs, 0);
        for (u32 k = 0; k < nCuts; ++k)
        {
            const u64 v = map->cut_positions[k] & ~u64(1);
            if (k && v <= (map->cut_positions[k - 1] & ~u64(1))) return fail(ISAAC_GPU_EINVAL, "cut_positions: ascending ReferencePosition values");
            if (refposContig(v) >= nContigs) return fail(ISAAC_GPU_EINVAL, "cut_positions: a cut on a contig that is not there");
            ++cutsOf[refposContig(v)];
        }
        for (u32 k = 0; k < nContigs; ++k) if (u64(map->bin_of_contig[k]) + cutsOf[k] + 1 >= nBins) return fail(ISAAC_GPU_EINVAL, "bin_of_contig: bins 0 .. n_bins - 2 (the last bin takes the templates without a position)");
    }
    for (u32 b = 0; b < nBins; ++b) { sizesOut[b].n_clusters = 0; sizesOut[b].n_cigar_words = 0; }
    if (!nClusters) return 0;
    if (!bcl || !fragments || !cigars) return fail(ISAAC_GPU_EINVAL, "bcl_dev, fragments_dev and cigar_dev are required");
    hipStream_t st = c->stream;
    const u32 nReads = c->params.n_reads, clusterLength = c->P.clusterLength;
    const u64 nEntries = u64(nClusters) * 2;
    c->binOfContig.reserve(std::max(nContigs, 1u)); c->binCuts.reserve(std::max(nCuts, 1u)); c->binKeys.reserve(nEntries); c->binKeysAlt.reserve(nEntries); c->binValues.reserve(nEntries); c->binValuesAlt.reserve(nEntries);
    c->binWords.reserve(nEntries + 1); c->binCounts.reserve(2 * size_t(nBins) + 2); c->binLayout.reserve(4 * size_t(nBins) + 1);
    if (nContigs) HIP_CHECK(hipMemcpyAsync(c->binOfContig.p, map->bin_of_contig, nContigs * 4, hipMemcpyHostToDevice, st));
    if (nCuts)
    {
        std::vector<u64> cuts(nCuts);
        for (u32 k = 0; k < nCuts; ++k) cuts[k] = map->cut_positions[k] & ~u64(1);
        HIP_CHECK(hipMemcpyAsync(c->binCuts.p, cuts.data(), nCuts * 8, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));              // `cuts` goes out of scope
    }
    HIP_CHECK(hipMemsetAsync(c->binCounts.p, 0, (2 * size_t(nBins) + 2) * 8, st));
    const FragmentRecord *records = reinterpret_cast<const FragmentRecord *>(fragments);
    BinMap m; m.binOfContig = c->binOfContig.p; m.nContigs = nContigs; m.cuts = c->binCuts.p; m.nCuts = nCuts; m.nBins = nBins;
    // an entry per (cluster, bin it has a stored record in): at most two; sorted by bin (stable: cluster order inside a bin)
    k_bin_entries<<<gridFor(nClusters, 256), 256, 0, st>>>(records, nClusters, nReads, m, c->binKeys.p, c->binValues.p);
    int keyBits = 1; while ((1u << keyBits) <= nBins) ++keyBits;             // keys 0 .. n_bins
    sortPairs(c, c->binKeys.p, c->binKeysAlt.p, c->binValues.p, c->binValuesAlt.p, nEntries, keyBits);
    // CIGAR words of every entry's cluster, their running sum in sorted order, and per bin the entries and the words
    k_bin_words<<<gridFor(nEntries, 256), 256, 0, st>>>(records, nReads, c->binKeysAlt.p, c->binValuesAlt.p, nEntries, nBins, c->binWords.p, reinterpret_cast<unsigned long long *>(c->binCounts.p));
    exclusiveSum(c, c->binWords.p, c->binWords.p, nEntries + 1);
    k_bin_layout<<<1, 1, 0, st>>>(c->binCounts.p, nBins, nReads, clusterLength, c->binLayout.p, c->binCounts.p + 2 * size_t(nBins));
    std::vector<u64> counts(2 * size_t(nBins) + 2);
    HIP_CHECK(hipMemcpyAsync(counts.data(), c->binCounts.p, counts.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (u32 b = 0; b < nBins; ++b) { sizesOut[b].n_clusters = counts[b]; sizesOut[b].n_cigar_words = counts[nBins + b]; }
    const u64 at = counts[2 * size_t(nBins)], firstEntry = counts[2 * size_t(nBins) + 1];
    if (nBytesOut) *nBytesOut = at;
    if (at > capacity) return fail(ISAAC_GPU_ECAPACITY, "out_dev is too small");
    if (!out) return fail(ISAAC_GPU_EINVAL, "out_dev is required");
    BinLayout layout; layout.firstEntry = c->binLayout.p; layout.bclAt = layout.firstEntry + nBins + 1; layout.recordsAt = layout.bclAt + nBins; layout.cigarsAt = layout.recordsAt + nBins;
    if (firstEntry) k_bin_gather<<<gridFor(firstEntry, 4), 256, 0, st>>>(bcl, records, cigars, nReads, clusterLength, c->binKeysAlt.p, c->binValuesAlt.p, firstEntry, c->binWords.p, layout, out);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
    ISAAC_CATCH
}
int isaac_gpu_bin_tile(isaac_gpu_ctx *c, const uint8_t *bcl, const isaac_fragment *fragments, const uint32_t *cigars, uint32_t nClusters, const uint32_t *binOfContig, uint32_t nContigs, uint32_t nBins,
                       uint8_t *out, uint64_t capacity, isaac_bin_size *sizesOut, uint64_t *nBytesOut)
{
    isaac_bin_map map; map.bin_of_contig = binOfContig; map.n_contigs = nContigs; map.cut_positions = nullptr; map.n_cuts = 0; map.n_bins = nBins;
    return isaac_gpu_bin_tile_map(c, bcl, fragments, cigars, nClusters, &map, out, capacity, sizesOut, nBytesOut);
}

uint64_t isaac_gpu_bgzf_store_bound(uint64_t nBytes) { return ((nBytes + BGZF_BLOCK_INPUT - 1) / BGZF_BLOCK_INPUT) * u64(BGZF_BLOCK_INPUT + BGZF_STORED_OVERHEAD) + 28; }
int isaac_gpu_bgzf_store(isaac_gpu_ctx *c, const uint8_t *data, uint64_t nBytes, int eofBlock, uint8_t *out, uint64_t capacity, uint64_t *nBytesOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nBytesOut) *nBytesOut = 0;
    if (nBytes && (!data || !out)) return fail(ISAAC_GPU_EINVAL, "data_dev and out_dev are required");
    const u64 nBlocks = (nBytes + BGZF_BLOCK_INPUT - 1) / BGZF_BLOCK_INPUT;
    const u64 total = nBytes + nBlocks * BGZF_STORED_OVERHEAD + (eofBlock ? 28 : 0);
    if (nBytesOut) *nBytesOut = total;
    if (total > capacity) return fail(ISAAC_GPU_ECAPACITY, "out_dev is too small (isaac_gpu_bgzf_store_bound)");
    if (nBlocks >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "at most 2^31 - 1 blocks per call");
    hipStream_t st = c->stream;
    if (!c->crcReady)
    {
        CrcConstants h; makeCrcConstants(h);
        c->crcConstants.reserve(1);
        HIP_CHECK(hipMemcpy(c->crcConstants.p, &h, sizeof(h), hipMemcpyHostToDevice));
        c->crcReady = true;
    }
    if (nBlocks)
    {
        ScopedTimer t(c, "bgzf_store");
        k_bgzf_store<<<u32(nBlocks), 256, 0, st>>>(data, nBytes, c->crcConstants.p, out);
        HIP_CHECK(hipGetLastError());
    }
    if (eofBlock)
    {
        static const unsigned char eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        HIP_CHECK(hipMemcpyAsync(out + total - 28, eof, 28, hipMemcpyHostToDevice, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return 0;
    ISAAC_CATCH
}

uint64_t isaac_gpu_bgzf_deflate_bound(uint64_t nBytes) { return ((nBytes + DEFLATE_BLOCK_INPUT - 1) / DEFLATE_BLOCK_INPUT) * u64(DEFLATE_BLOCK_INPUT + 31) + 28; }
int isaac_gpu_bgzf_deflate(isaac_gpu_ctx *c, const uint8_t *data, uint64_t nBytes, int eofBlock, uint8_t *out, uint64_t capacity, uint64_t *nBytesOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (nBytesOut) *nBytesOut = 0;
    if (nBytes && (!data || !out)) return fail(ISAAC_GPU_EINVAL, "data_dev and out_dev are required");
    const u64 nBlocks = (nBytes + DEFLATE_BLOCK_INPUT - 1) / DEFLATE_BLOCK_INPUT;
    if (nBlocks >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "at most 2^31 - 1 blocks per call");
    hipStream_t st = c->stream;
    if (!c->crcReady)
    {
        CrcConstants h; makeCrcConstants(h);
        c->crcConstants.reserve(1);
        HIP_CHECK(hipMemcpy(c->crcConstants.p, &h, sizeof(h), hipMemcpyHostToDevice));
        c->crcReady = true;
    }
    u64 total = 0;
    if (nBlocks)
    {
        ScopedTimer t(c, "bgzf_deflate");
        // 1. the call's Huffman tables from the symbol counts of a sample of its blocks (every block when there are few)
        const u64 sampleMax = 2048, stride = (nBlocks + sampleMax - 1) / sampleMax, nSample = (nBlocks + stride - 1) / stride;
        c->deflateCounts.reserve(DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS); c->deflateTables.reserve(1);
        HIP_CHECK(hipMemsetAsync(c->deflateCounts.p, 0, sizeof(u64) * (DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS), st));
        k_deflate_blocks<<<u32(nSample), 64, 0, st>>>(data, nBytes, 0, stride, nBlocks, nullptr, c->crcConstants.p, 1, reinterpret_cast<unsigned long long *>(c->deflateCounts.p), nullptr, nullptr);
        HIP_CHECK(hipGetLastError());
        u64 counts[DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS];
        HIP_CHECK(hipMemcpyAsync(counts, c->deflateCounts.p, sizeof(counts), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        DeflateTables tables;
        if (!makeDeflateTables(counts, counts + DEFLATE_LITLEN_SYMBOLS, tables)) return fail(ISAAC_GPU_EHIP, "the deflate block header does not fit its buffer");
        HIP_CHECK(hipMemcpyAsync(c->deflateTables.p, &tables, sizeof(tables), hipMemcpyHostToDevice, st));
        // 2. the blocks, a slice at a time: every block into its own slot, then closed up behind what is there already
        const u64 slice = 16384;
        c->deflateStaging.reserve(size_t(std::min(nBlocks, slice)) * DEFLATE_SLOT); c->deflateSizes.reserve(nBlocks); c->deflateOffsets.reserve(std::min(nBlocks, slice));
        for (u64 first = 0; first < nBlocks; first += slice)
        {
            const u64 n = std::min(slice, nBlocks - first);
            // (the kernels address slots and sizes by absolute block number: the slice's buffers are offset accordingly)
            u8 *staging = c->deflateStaging.p - first * DEFLATE_SLOT;
            k_deflate_blocks<<<u32(n), 64, 0, st>>>(data, nBytes, first, 1, nBlocks, c->deflateTables.p, c->crcConstants.p, 0, nullptr, staging, c->deflateSizes.p);
            HIP_CHECK(hipGetLastError());
            k_widen_sizes<<<gridFor(n, 256), 256, 0, st>>>(c->deflateSizes.p + first, n, c->deflateOffsets.p);
            u64 lastSize = 0, lastOffset = 0;
            HIP_CHECK(hipMemcpyAsync(&lastSize, c->deflateOffsets.p + n - 1, 8, hipMemcpyDeviceToHost, st));
            exclusiveSum(c, c->deflateOffsets.p, c->deflateOffsets.p, n);
            HIP_CHECK(hipMemcpyAsync(&lastOffset, c->deflateOffsets.p + n - 1, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const u64 sliceBytes = lastOffset + lastSize;
            if (total + sliceBytes + (eofBlock ? 28 : 0) > capacity)
            {
                if (nBytesOut) *nBytesOut = isaac_gpu_bgzf_deflate_bound(nBytes);
                return fail(ISAAC_GPU_ECAPACITY, "out_dev is too small (isaac_gpu_bgzf_deflate_bound is always enough)");
            }
            k_deflate_gather<<<u32(n), 256, 0, st>>>(c->deflateStaging.p, c->deflateSizes.p + first, c->deflateOffsets.p, n, out + total);
            HIP_CHECK(hipGetLastError());
            total += sliceBytes;
        }
    }
    if (eofBlock)
    {
        static const unsigned char eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        if (total + 28 > capacity) return fail(ISAAC_GPU_ECAPACITY, "out_dev is too small");
        HIP_CHECK(hipMemcpyAsync(out + total, eof, 28, hipMemcpyHostToDevice, st));
        total += 28;
    }
    HIP_CHECK(hipStreamSynchronize(st));
    if (nBytesOut) *nBytesOut = total;
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_bsw_batch(isaac_gpu_ctx *c, int match, int mismatch, int gapOpen, int gapExtend, const char *sequences, const isaac_bsw_job *jobs, uint32_t nJobs,
                        uint32_t maxQueryLength, isaac_bsw_result *results)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (!nJobs) return 0;
    if (!maxQueryLength || maxQueryLength > 512) return fail(ISAAC_GPU_EINVAL, "query lengths 1..512 are supported");
    const int maxScore = std::max(std::max(std::abs(match), std::abs(mismatch)), std::max(std::abs(gapOpen), std::abs(gapExtend)));
    if (int(maxQueryLength) * maxScore >= std::abs(-32768 + gapOpen)) return fail(ISAAC_GPU_EINVAL, "BandedSmithWaterman: unsupported read length for these scores");
    const size_t lds = size_t(BSW_BLOCK / BSW_GROUP_LANES) * bswGroupLdsBytes(maxQueryLength);
    ScopedTimer t(c, "bsw");
    k_bsw_batch<<<gridFor(nJobs, BSW_BLOCK / BSW_GROUP_LANES), BSW_BLOCK, lds, c->stream>>>(match, mismatch, gapOpen, gapExtend, sequences, jobs, nJobs, maxQueryLength, results);
    HIP_CHECK(hipGetLastError());
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_fastq_to_bcl(isaac_gpu_ctx *c, const char *fastq, uint64_t nBytes, uint32_t readIndex, int allowVariableLength, int final, uint8_t *bcl,
                           uint32_t maxClusters, uint32_t *nClustersOut, uint64_t *consumedOut, uint64_t *errorOffsetOut)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    if (!nClustersOut || !consumedOut) return fail(ISAAC_GPU_EINVAL, "n_clusters_out and consumed_bytes_out are required");
    *nClustersOut = 0; *consumedOut = 0; if (errorOffsetOut) *errorOffsetOut = 0;
    if (readIndex >= c->P.nReads) return fail(ISAAC_GPU_EINVAL, "read_index");
    if (nBytes >= (u64(1) << 31)) return fail(ISAAC_GPU_EINVAL, "pieces of FASTQ text below 2 GiB per call");
    if (!nBytes) return 0;
    if (!fastq || !bcl) return fail(ISAAC_GPU_EINVAL, "fastq_dev and bcl_dev are required");
    hipStream_t st = c->stream;
    const u32 readLength = c->P.readLength[readIndex];
    // 1. lines
    DevBuf<u8> &isStart = c->fqIsStart; isStart.reserve(nBytes);
    DevBuf<u64> &lineStart = c->fqLineStart; lineStart.reserve(nBytes / 2 + 2);
    DevBuf<int> &nSelected = c->fqSelected; nSelected.reserve(1);
    k_fq_line_starts<<<gridFor(nBytes, 256), 256, 0, st>>>(fastq, nBytes, isStart.p);
    {
        checkCount(nBytes, "FASTQ text of one call");
        rocprim::counting_iterator<u64> positions(0);
        size_t bytes = 0;
        HIP_CHECK(rocprim::select(nullptr, bytes, positions, isStart.p, lineStart.p, nSelected.p, size_t(nBytes), st));
        c->cubTemp.reserve(bytes + 16);
        HIP_CHECK(rocprim::select(c->cubTemp.p, bytes, positions, isStart.p, lineStart.p, nSelected.p, size_t(nBytes), st));
    }
    int nLinesHost = 0;
    HIP_CHECK(hipMemcpyAsync(&nLinesHost, nSelected.p, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const u32 nLines = u32(nLinesHost);
    if (!nLines) { *consumedOut = nBytes; return 0; }        // nothing but newlines
    // 2. the role of every line
    DevBuf<u64> &lineEnd = c->fqLineEnd; DevBuf<u32> &lineMap = c->fqLineMap, &mapBefore = c->fqMapBefore, &isHeader = c->fqIsHeader, &recordIndex = c->fqRecordIndex;
    lineEnd.reserve(nLines); lineMap.reserve(nLines); mapBefore.reserve(nLines); isHeader.reserve(nLines); recordIndex.reserve(nLines);
    k_fq_lines<<<gridFor(nLines, 256), 256, 0, st>>>(fastq, nBytes, lineStart.p, nLines, lineEnd.p, lineMap.p);
    {
        size_t bytes = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, lineMap.p, mapBefore.p, FQ_IDENTITY, size_t(nLines), FqCompose(), st));
        c->cubTemp.reserve(bytes + 16);
        HIP_CHECK(rocprim::exclusive_scan(c->cubTemp.p, bytes, lineMap.p, mapBefore.p, FQ_IDENTITY, size_t(nLines), FqCompose(), st));
    }
    k_fq_headers<<<gridFor(nLines, 256), 256, 0, st>>>(mapBefore.p, nLines, isHeader.p);
    exclusiveSum(c, isHeader.p, recordIndex.p, nLines);
    u32 lastIndex = 0, lastFlag = 0;
    HIP_CHECK(hipMemcpyAsync(&lastIndex, recordIndex.p + nLines - 1, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(&lastFlag, isHeader.p + nLines - 1, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const u32 nRecords = lastIndex + lastFlag;
    const u32 nTried = std::min(nRecords, maxClusters);
    if (!nTried) { *consumedOut = nRecords ? 0 : nBytes; return 0; }
    // 3. the records
    DevBuf<FqRecord> &records = c->fqRecords; records.reserve(nTried);
    DevBuf<u32> &firstBad = c->fqFirstBad; firstBad.reserve(1);
    HIP_CHECK(hipMemsetAsync(firstBad.p, 0xff, 4, st));
    {
        ScopedTimer t(c, "fastq_to_bcl");
        k_fq_records<<<gridFor(nLines, 256), 256, 0, st>>>(fastq, nBytes, final, allowVariableLength, readLength, lineStart.p, lineEnd.p, lineMap.p, isHeader.p, recordIndex.p, nLines,
                                                            bcl + c->P.readOffset[readIndex], c->P.clusterLength, maxClusters, records.p);
        HIP_CHECK(hipGetLastError());
    }
    k_fq_first_bad<<<gridFor(nTried, 256), 256, 0, st>>>(records.p, nTried, firstBad.p);
    u32 bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, firstBad.p, 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const u32 nGood = std::min(bad, nTried);
    *nClustersOut = nGood;
    FqRecord lastGood, firstBadRecord;
    if (nGood) HIP_CHECK(hipMemcpy(&lastGood, records.p + nGood - 1, sizeof(FqRecord), hipMemcpyDeviceToHost));
    *consumedOut = (nGood == nRecords) ? nBytes : (nGood ? lastGood.recordEnd : 0);
    if (bad < nTried)
    {
        HIP_CHECK(hipMemcpy(&firstBadRecord, records.p + bad, sizeof(FqRecord), hipMemcpyDeviceToHost));
        if (FQ_INCOMPLETE != firstBadRecord.status)
        {
            if (errorOffsetOut) *errorOffsetOut = firstBadRecord.errorOffset;
            return FQ_BAD_LENGTH == firstBadRecord.status ? fail(ISAAC_GPU_EREADLEN, "FASTQ read length is different from expected (common::IoException)")
                                                          : fail(ISAAC_GPU_EFORMAT, "malformed FASTQ record (io::FastqFormatException)");
        }
    }
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_get_counters(isaac_gpu_ctx *c, isaac_counters *out)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    static_assert(sizeof(isaac_counters) == sizeof(Counters), "counter layouts");
    std::vector<Counters> shards(COUNTER_SHARDS);
    HIP_CHECK(hipStreamSynchronize(c->stream));     // kernels in flight still count
    HIP_CHECK(hipMemcpy(shards.data(), c->counters.p, COUNTER_SHARDS * sizeof(Counters), hipMemcpyDeviceToHost));
    u64 *sum = reinterpret_cast<u64 *>(out);
    for (u32 f = 0; f < sizeof(Counters) / sizeof(u64); ++f)
    {
        sum[f] = 0;
        for (u32 i = 0; i < COUNTER_SHARDS; ++i) sum[f] += reinterpret_cast<const u64 *>(&shards[i])[f];
    }
    return 0;
    ISAAC_CATCH
}

int isaac_gpu_kernel_time_ms(isaac_gpu_ctx *c, const char *kernel, double *avgMs, uint64_t *launches)
{
    resolveTimers(c);
    const auto it = c->timers.find(kernel);
    if (it == c->timers.end() || !it->second.launches) { if (avgMs) *avgMs = 0; if (launches) *launches = 0; return 0; }
    if (avgMs) *avgMs = it->second.ms / double(it->second.launches);
    if (launches) *launches = it->second.launches;
    return 0;
}
int isaac_gpu_reset_timers(isaac_gpu_ctx *c)
{
    ISAAC_TRY
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    resolveTimers(c);
    c->timers.clear();
    HIP_CHECK(hipMemset(c->counters.p, 0, COUNTER_SHARDS * sizeof(Counters)));
    return 0;
    ISAAC_CATCH
}

} // extern "C"
