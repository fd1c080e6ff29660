// kernels_fragment.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"
#include "bsw_kernel.h"
#include "fragment_lean.h"

__device__ inline void emitGappedJobs(const DevParams &P, const ClusterFragments &f, u32 cl, bool withGaps, const GappedBuffers &gb)
{
    u32 base = 0;
    const u32 n = countGappedJobs(f, withGaps);
    if (n)
    {
        base = atomicAdd(gb.counter, n);
        if (base + n > gb.cap) base = 0xffffffffu;   // k_finish_fragments runs this cluster's retries itself
        else writeGappedJobs(P, f, cl, gb.jobs + base);
    }
    gb.base[cl] = base;
}


// The general form of a step runs over a list of clusters (count on the device).  These are the clusters of repeat families -- dozens of
// matches and candidates each -- and what a thread does for one is a chain of instruction-exact sorts: a latency chain that only faster memory
// shortens.  A workgroup (one wavefront) therefore takes GENERAL_LANES entries at a time, one per lane of its first GENERAL_LANES lanes (the
// others idle along: the wave-level scans count them as clusters without work), so that every cluster's index arrays and sort keys fit in LDS.
static const u32 GENERAL_LANES = 16;
#define ISAAC_LIST_LOOP(listCount) for (u32 listBase = blockIdx.x * GENERAL_LANES, listN = *(listCount); listBase < listN; listBase += gridDim.x * GENERAL_LANES)
// the lane's work area in LDS: list order, sort keys of CAND_CAP candidates (interleaved by lane)
#define ISAAC_GENERAL_WORK(work, tflagsArena)                                                          \
    __shared__ u8 generalOrder[GENERAL_LANES][CAND_CAP];                                              \
    __shared__ u64 generalKeysA[CAND_CAP * GENERAL_LANES];                                            \
    __shared__ u64 generalKeysB[CAND_CAP * GENERAL_LANES];                                            \
    const u32 generalLane = threadIdx.x & (GENERAL_LANES - 1);                                         \
    LeanKeyArea generalKeys; generalKeys.a = generalKeysA + generalLane; generalKeys.b = generalKeysB + generalLane; generalKeys.stride = GENERAL_LANES; \
    FragmentWork work; work.order = generalOrder[generalLane]; work.matchOrder = 0; work.keys = &generalKeys; work.keyCap = CAND_CAP;    \
    work.tflags = (tflagsArena) ? (tflagsArena) + (size_t(blockIdx.x) * GENERAL_LANES + generalLane) * 3 * 512 : 0;

// a wavefront's clusters for the general form: one slot of the list per lane that has one
__device__ inline void pushGeneral(bool mine, u32 cl, u32 *list, u32 *count)
{
    const u64 ballot = __ballot(mine);
    if (!ballot) return;
    const u32 lane = threadIdx.x & 63;
    u32 base = 0;
    if (lane == u32(__ffsll((unsigned long long)ballot)) - 1) base = atomicAdd(count, u32(__popcll(ballot)));
    base = __shfl(base, __ffsll((unsigned long long)ballot) - 1, 64);
    if (mine) list[base + u32(__popcll(ballot & ((u64(1) << lane) - 1)))] = cl;
}

// the per-thread key area of the lean steps (fragment_lean.h): two 64-bit words per list entry, interleaved by lane
#define ISAAC_LEAN_KEY_AREA(name, entries)                                            \
    __shared__ u64 leanKeysA[(entries) * 64];                                         \
    __shared__ u64 leanKeysB[(entries) * 64];                                         \
    LeanKeyArea name; name.a = leanKeysA + (threadIdx.x & 63); name.b = leanKeysB + (threadIdx.x & 63); name.stride = 64;

// one entry per candidate in the flat list k_align_candidates works through; the list space of a wave is taken with one atomic
__device__ inline void listCandidates(ClusterFragments &f, u32 cl, u32 n, const AlignList &al)
{
    u32 incl = n;
    for (u32 o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += v; }
    const u32 total = __shfl(incl, 63, 64);
    u32 base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(al.counter, total);
    base = __shfl(base, 63, 64);
    if (n)
    {
        u32 at = base + incl - n;
        if (base + total > al.cap)
        {   // no room: the cluster aligns its own candidates later; what the wave took of the list is marked unused
            f.flags |= CLUSTER_ALIGN_PENDING;
            for (u32 k = 0; k < n; ++k) if (at + k < al.cap) al.entries[at + k] = 0xffffffffu;
        }
        else for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) al.entries[at++] = (cl << 8) | (r << 7) | i;
    }
}

// Fragment stage, step 1: matches -> candidate positions.  Clusters with up to BUILD_LEAN_MAX matches (all but a per cent) are done
// here on keys in LDS (fragment_lean.h: leanBuildCandidates); the others are listed for k_build_fragments_general.
__global__ __launch_bounds__(64) void k_build_fragments(DevParams P, const u8 *__restrict__ bcl, u32 clusterBase, u32 nChunk, const Match *__restrict__ matches, const u64 *__restrict__ offsets,
                                                        int trim, ClusterPools pools, AlignList al, u32 *generalList, u32 *generalCount)
{
    ISAAC_LEAN_KEY_AREA(keys, BUILD_LEAN_MAX)
    const u32 cl = blockIdx.x * blockDim.x + threadIdx.x;
    u32 n = 0;
    ClusterFragments f;
    bool general = false;
    if (cl < nChunk)
    {
        // the cluster's slots: one per seed match, at the offset of its first match in the chunk
        const u64 chunkBegin = offsets[clusterBase], begin = offsets[clusterBase + cl], end = offsets[clusterBase + cl + 1];
        const u32 first = u32(begin - chunkBegin);
        const u32 nMatches = u32(end - begin);
        general = nMatches > BUILD_LEAN_MAX || u64(first) + nMatches > pools.candCap;      // (a pool that is too small shows as CLUSTER_OVERFLOW there)
        if (!general)
        {
            f = clusterViewNew(first, nMatches, pools.cands, pools.cigars);
            leanBuildCandidates(P, bcl + u64(clusterBase + cl) * P.clusterLength, matches + begin, nMatches, trim != 0, f, keys);
            n = f.nCands[0] + f.nCands[1];
        }
    }
    pushGeneral(general, cl, generalList, generalCount);
    listCandidates(f, cl, n, al);
    if (cl < nChunk && !general) clusterViewStore(f, pools.cands, pools.meta[cl]);
}

// step 1 in its general form for the listed clusters: on keys in LDS (fragment_lean.h: keyedBuildCandidates) when the cluster's matches fit
// there -- GENERAL_STAGE_MATCHES = 2 strands x 8 seeds x repeat threshold 10 -- else where they lie (aligner.h: buildCandidates)
__global__ __launch_bounds__(64) void k_build_fragments_general(DevParams P, const u8 *bcl, u32 clusterBase, const Match *matches, const u64 *offsets,
                                                        int trim, u8 *matchOrderArena, u8 *orderArena, ClusterPools pools, AlignList al, const u32 *list, const u32 *listCount)
{
    __shared__ u64 keysA[GENERAL_STAGE_MATCHES * GENERAL_LANES];
    __shared__ u64 keysB[GENERAL_STAGE_MATCHES * GENERAL_LANES];
    __shared__ u8 matchOrder[GENERAL_LANES][GENERAL_STAGE_MATCHES];
    __shared__ u8 candOrder[GENERAL_LANES][GENERAL_STAGE_MATCHES];
    const u32 generalLane = threadIdx.x & (GENERAL_LANES - 1);
    LeanKeyArea keys; keys.a = keysA + generalLane; keys.b = keysB + generalLane; keys.stride = GENERAL_LANES;
    FragmentWork work; work.order = orderArena + (size_t(blockIdx.x) * GENERAL_LANES + generalLane) * CAND_CAP; work.tflags = 0; work.keys = 0; work.keyCap = 0;
    work.matchOrder = matchOrderArena + (size_t(blockIdx.x) * GENERAL_LANES + generalLane) * MATCH_CAP_MAX;
    ISAAC_LIST_LOOP(listCount)
    {
        const u32 t = listBase + threadIdx.x;
        const bool active = threadIdx.x < GENERAL_LANES && t < listN;
        u32 cl = 0, n = 0;
        ClusterFragments f;
        if (active)
        {
            cl = list[t];
            const u64 chunkBegin = offsets[clusterBase], begin = offsets[clusterBase + cl], end = offsets[clusterBase + cl + 1];
            const u32 first = u32(begin - chunkBegin);
            const u32 nMatches = u32(end - begin);
            const u32 cap = (u64(first) + nMatches <= pools.candCap) ? nMatches : 0u;     // a pool that is too small shows as CLUSTER_OVERFLOW
            if (!cap && nMatches) atomicOr(pools.shortFlag, 1u);
            f = clusterViewNew(first, cap, pools.cands, pools.cigars);
            const u8 *clusterBcl = bcl + u64(clusterBase + cl) * P.clusterLength;
            if (cap && nMatches <= GENERAL_STAGE_MATCHES) keyedBuildCandidates(P, clusterBcl, matches + begin, nMatches, trim != 0, f, keys, matchOrder[generalLane], candOrder[generalLane]);
            else buildCandidates(P, clusterBcl, matches + begin, nMatches, trim != 0, work, f, nullptr);
            n = f.nCands[0] + f.nCands[1];
        }
        listCandidates(f, cl, n, al);
        if (active) clusterViewStore(f, pools.cands, pools.meta[cl]);
    }
}

// between steps 1 and 2, with sequencing adapters only (--default-adapters): where each read's adapter lies on either strand, decided by the first candidate
// of that strand in the consolidated list as built (FragmentSequencingAdapterClipper::checkInitStrand, FragmentBuilder.cpp:164-174).  A thread per cluster,
// read and strand; the four words per cluster are what every later alignment of the cluster clips by.
__global__ __launch_bounds__(256) void k_adapter_ranges(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, u32 nChunk, ClusterPools pools)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 cl = t >> 2, r = (t >> 1) & 1, strand = t & 1;
    if (cl >= nChunk || r >= P.nReads) return;
    const ClusterMeta meta = pools.meta[cl];
    if (!meta.cap) return;                                   // (a cluster without a match owns no slot: nothing of it is ever aligned)
    const ClusterFragments f = clusterView(meta, pools.cands, pools.cigars);
    clusterInitAdapterRanges(P, R, bcl + u64(clusterBase + cl) * P.clusterLength, f, r, strand);
}

// step 2: UngappedAligner::alignUngapped, one candidate per thread
__global__ __launch_bounds__(256) void k_align_candidates(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, ClusterPools pools, AlignList al, Counters *counters)
{
    ISAAC_STAGE_SCAN_TABLES(Rg, R)
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = imin(*al.counter, al.cap);
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    {
        const u32 e = al.entries[j], cl = e >> 8;
        if (0xffffffffu == e) continue;
        ClusterFragments f = clusterView(pools.meta[cl], pools.cands, pools.cigars);      // the candidate and its three cigar words are written in place; the cluster's state is not touched
        alignCandidate(P, R, bcl + u64(clusterBase + cl) * P.clusterLength, f, (e >> 7) & 1, e & 127, local);
    }
    flushCounters(local, counters);
}

// step 3: consolidation and the single-indel stage (finishCandidates), then either the cluster's gapped problems or, for the
// 3-4 % of clusters with a candidate pair for the single-indel detector, an entry for k_indel_fragments: inside this kernel
// nearly every wave would hold one such lane and wait for it.  Lists of up to FINISH_LEAN_MAX candidates are done on keys in LDS
// (fragment_lean.h: leanFinishCandidates); clusters with a longer one, or whose candidates are still to be aligned, are listed for
// k_finish_candidates_general.
__global__ __launch_bounds__(64) void k_finish_candidates(DevParams P, u32 nChunk, int withGaps, u32 *indelList, u32 *indelCount, ClusterPools pools, GappedBuffers gb,
                                                          const u32 *__restrict__ order, u32 *generalList, u32 *generalCount)
{
    ISAAC_LEAN_KEY_AREA(keys, FINISH_LEAN_MAX)
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    bool general = false; u32 cl = 0;
    if (t < nChunk)
    {
        cl = order ? order[t] : t;
        ClusterFragments f = clusterView(pools.meta[cl], pools.cands, pools.cigars);
        general = f.nCands[0] > FINISH_LEAN_MAX || f.nCands[1] > FINISH_LEAN_MAX || (f.flags & CLUSTER_ALIGN_PENDING);
        if (!general)
        {
            leanFinishCandidates(P, f, keys);
            if (clusterSimpleIndelsPending(f)) indelList[atomicAdd(indelCount, 1u)] = cl;
            else emitGappedJobs(P, f, cl, withGaps != 0, gb);
            clusterViewStore(f, pools.cands, pools.meta[cl]);
        }
    }
    pushGeneral(general, cl, generalList, generalCount);
}

__global__ __launch_bounds__(64) void k_finish_candidates_general(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase,
                                                        int withGaps, u32 *indelList, u32 *indelCount, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *list, const u32 *listCount)
{
    ISAAC_STAGE_QUALITY_TABLES(Rg, R)
    ISAAC_GENERAL_WORK(work, (u32 *)nullptr)
    Counters local; memset(&local, 0, sizeof(local));
    ISAAC_LIST_LOOP(listCount)
    {
        const u32 t = listBase + threadIdx.x;
        if (threadIdx.x >= GENERAL_LANES || t >= listN) continue;
        const u32 cl = list[t];
        ClusterFragments f = clusterView(pools.meta[cl], pools.cands, pools.cigars);
        const u8 *clusterBcl = bcl + u64(clusterBase + cl) * P.clusterLength;
        if (f.flags & CLUSTER_ALIGN_PENDING)
        {
            f.flags &= ~u32(CLUSTER_ALIGN_PENDING);
            for (u32 r = 0; r < P.nReads; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) alignCandidate(P, R, clusterBcl, f, r, i, local);
        }
        finishCandidates(P, R, clusterBcl, work, f, local, true);
        if (clusterSimpleIndelsPending(f)) indelList[atomicAdd(indelCount, 1u)] = cl;
        else emitGappedJobs(P, f, cl, withGaps != 0, gb);
        clusterViewStore(f, pools.cands, pools.meta[cl]);
    }
    flushCounters(local, counters);
}

// The deferred single-indel stage (SimpleIndelAligner) for the clusters k_build_fragments listed, then their gapped problems.
// One wave per cluster, every lane executing the same statements (as in k_select_heavy): the detector is a chain of dependent
// byte loads, and 64 different clusters per wave would spread them over more cache lines than the L1 holds.
__global__ __launch_bounds__(64) void k_indel_fragments(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, int withGaps, const u32 *indelList, const u32 *indelCount,
                                                        ClusterPools pools, GappedBuffers gb, Counters *counters)
{
    ISAAC_STAGE_QUALITY_TABLES(Rg, R)
    __shared__ u8 listOrder[CAND_CAP];                  // every lane writes the same values
    FragmentWork work; work.order = listOrder; work.matchOrder = 0; work.tflags = 0; work.keys = 0; work.keyCap = 0;
    __shared__ __align__(16) u8 stageBcl[512];
    __shared__ __align__(16) char stageWindow[1536];
    IndelStage stage; stage.bcl = stageBcl; stage.bclCap = sizeof(stageBcl); stage.window = stageWindow; stage.windowCap = sizeof(stageWindow); stage.lane = threadIdx.x;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *indelCount;
    for (u32 t = blockIdx.x; t < n; t += gridDim.x)
    {
        const u32 cl = indelList[t];
        ClusterFragments f = clusterView(pools.meta[cl], pools.cands, pools.cigars);       // the same view in every lane
        // an accepted single indel writes a CIGAR of up to five words; there is at most one per neighbouring pair of a list
        const u32 need = 5 * (f.nCands[0] + f.nCands[1]);
        u32 at = 0;
        if (0 == threadIdx.x) at = 3 * pools.candCap + atomicAdd(pools.cigarNext, need);
        at = __shfl(at, 0, 64);
        clusterCigarExtra(f, pools.cigars, at, need, pools.cigarCap);
        clusterFinishSimpleIndels(P, R, bcl, clusterBase + cl, work, f, local, &stage);
        __syncthreads();
        // a cluster that ran out of CIGAR words after the arena had none left for it: the call is repeated with a larger arena (selectFromSource)
        if (0 == threadIdx.x && u64(at) + need > pools.cigarCap && (f.flags & CLUSTER_OVERFLOW)) atomicOr(pools.shortFlag, 2u);
        if (0 == threadIdx.x) { emitGappedJobs(P, f, cl, withGaps != 0, gb); clusterViewStore(f, pools.cands, pools.meta[cl]); }
    }
    if (0 != threadIdx.x) memset(&local, 0, sizeof(local));   // every lane counted the same events
    flushCounters(local, counters);
}

// the cluster's share of the cigar arena for the gapped alignments that may be accepted: one bump of the arena's counter per wave
// true: the arena had no room left (the cluster keeps what is left of its own three words per candidate)
__device__ inline bool reserveGappedCigars(ClusterFragments &f, bool active, u32 need, const ClusterPools &pools)
{
    u32 incl = need;
    for (u32 o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += v; }
    const u32 total = __shfl(incl, 63, 64);
    u32 base = 0;
    if ((threadIdx.x & 63) == 63 && total) base = atomicAdd(pools.cigarNext, total);
    base = __shfl(base, 63, 64);
    if (!active || !need) return false;
    clusterCigarExtra(f, pools.cigars, 3 * pools.candCap + base + incl - need, need, pools.cigarCap);
    return u64(3) * pools.candCap + base + incl > pools.cigarCap;
}
// a cluster that ran out of CIGAR words after the arena had none left for it: the call is repeated with a larger arena (selectFromSource)
__device__ inline void arenaShort(bool reservationFailed, const ClusterFragments &f, const ClusterPools &pools) { if (reservationFailed && (f.flags & CLUSTER_OVERFLOW)) atomicOr(pools.shortFlag, 2u); }

// step 5: the accept rule for the gapped alignments and the final consolidation.  Lists of up to FINISH_LEAN_MAX candidates on keys in LDS
// (fragment_lean.h: leanFinishFragments); clusters with a longer one, or whose gapped problems found no room in the flat pass, are listed
// for k_finish_fragments_general.
__global__ __launch_bounds__(64) void k_finish_fragments(DevParams P, u32 nChunk, int withGaps, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *__restrict__ order,
                                                         u32 *generalList, u32 *generalCount)
{
    ISAAC_LEAN_KEY_AREA(keys, FINISH_LEAN_MAX)
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = slot < nChunk ? (order ? order[slot] : slot) : nChunk;
    u32 bswJobs = 0, bswAccepted = 0, candidates = 0;
    ClusterFragments f;
    const GappedResult *res = nullptr;
    u32 need = 0;
    bool general = false;
    if (t < nChunk)
    {
        f = clusterView(pools.meta[t], pools.cands, pools.cigars);
        general = f.nCands[0] > FINISH_LEAN_MAX || f.nCands[1] > FINISH_LEAN_MAX;
        if (!general)
        {
            const u32 nJobs = countGappedJobs(f, withGaps != 0);
            if (nJobs && gb.base[t] == 0xffffffffu) general = true;
            else if (nJobs)
            {
                res = gb.results + gb.base[t];
                for (u32 k = 0; k < nJobs; ++k) { const u32 w = res[k].nCigar; need += (0xffffffffu == w) ? 0u : w; }
            }
        }
    }
    pushGeneral(general, t, generalList, generalCount);
    const bool active = t < nChunk && !general;
    const bool arenaFull = reserveGappedCigars(f, active, need, pools);
    if (active)
    {
        leanFinishFragments(P, f, res, keys, bswJobs, bswAccepted, candidates);
        arenaShort(arenaFull, f, pools);
        if (f.flags & CLUSTER_OVERFLOW) flushCounter(&Counters::overflowClusters, 1, counters);
        clusterViewStore(f, pools.cands, pools.meta[t]);
    }
    flushCounter(&Counters::bswJobs, bswJobs, counters);
    flushCounter(&Counters::bswAccepted, bswAccepted, counters);
    flushCounter(&Counters::candidates, candidates, counters);
}

__global__ __launch_bounds__(64) void k_finish_fragments_general(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, int withGaps,
                                                         u32 *tflagsArena, ClusterPools pools, GappedBuffers gb, Counters *counters, const u32 *list, const u32 *listCount)
{
    ISAAC_GENERAL_WORK(work, tflagsArena)
    Counters local; memset(&local, 0, sizeof(local));
    ISAAC_LIST_LOOP(listCount)
    {
        const u32 slot = listBase + threadIdx.x;
        const bool active = threadIdx.x < GENERAL_LANES && slot < listN;
        const u32 t = active ? list[slot] : 0;
        ClusterFragments f;
        const GappedResult *res = nullptr;
        u32 need = 0;
        if (active)
        {
            f = clusterView(pools.meta[t], pools.cands, pools.cigars);
            const u32 nJobs = countGappedJobs(f, withGaps != 0);
            res = (nJobs && gb.base[t] != 0xffffffffu) ? gb.results + gb.base[t] : nullptr;
            // room for the CIGARs of the gapped alignments that may be accepted (all of them at most; 40 words each when they are still to run)
            if (res) for (u32 k = 0; k < nJobs; ++k) { const u32 w = res[k].nCigar; need += (0xffffffffu == w) ? 0u : w; }
            else need = 40 * nJobs;
        }
        const bool arenaFull = reserveGappedCigars(f, active, need, pools);
        if (active)
        {
            clusterFinishFragments(P, R, bcl, clusterBase + t, withGaps != 0, res, work, f, local);
            arenaShort(arenaFull, f, pools);
            clusterViewStore(f, pools.cands, pools.meta[t]);
        }
    }
    flushCounters(local, counters);
}
