// kernels_sums.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"

__device__ inline SumInputs sumInputs(const RescueBuffers &rb, u32 t, const GappedBuffers &gb)
{
    SumInputs in; in.jobs = rb.jobs + rb.jobBase[t]; in.nJobs = rb.jobCount[t]; in.shadowCands = rb.shadowCands; in.candRank = rb.candRank; in.gappedResults = gb.results; in.gappedJobs = gb.jobs; in.shadowCigars = rb.shadowCigars;
    return in;
}
__device__ inline void markResidual(const SumsBuffers &sb, u32 t) { sb.residualFlag[t] = 1; sb.residualList[atomicAdd(sb.residualCount, 1u)] = t; }
// the same when several workgroups (one per list of the cluster) may come to that conclusion: the first one lists the cluster
__device__ inline bool markResidualOnce(const SumsBuffers &sb, u32 t)
{
    u32 *word = reinterpret_cast<u32 *>(sb.residualFlag + (t & ~3u));
    const u32 bit = 1u << (8 * (t & 3u));
    if (atomicOr(word, bit) & bit) return false;
    sb.residualList[atomicAdd(sb.residualCount, 1u)] = t;
    return true;
}
// The cluster's problem records in LDS for the workgroup tiers: clusterSums walks them several times, a field at a time, every walk a chain of
// round trips to the L2 (ten problems: 30 us of a list's 60).  They are only read there (the problems were finished by the first tier).
static const u32 SUMS_STAGED_JOBS = 64;
__device__ inline SumInputs stagedInputs(const RescueBuffers &rb, u32 t, const GappedBuffers &gb, RescueJob *staged)
{
    SumInputs in = sumInputs(rb, t, gb);
    __syncthreads();                                   // the last list's readers are done with the copy
    if (in.nJobs <= SUMS_STAGED_JOBS)
    {
        const u32 *from = reinterpret_cast<const u32 *>(in.jobs); u32 *to = reinterpret_cast<u32 *>(staged);
        for (u32 w = threadIdx.x; w < in.nJobs * u32(sizeof(RescueJob) / 4); w += blockDim.x) to[w] = from[w];
        in.jobs = staged;
    }
    __syncthreads();
    return in;
}
// the next (cluster, list) item of a tier, handed out as the workgroups become free, the pair lists -- the longest -- first: a fixed stride gave
// the workgroups that began with a long list a second one
__device__ inline u32 nextItem(u32 *counter, u32 *slot)
{
    __syncthreads();                                   // everybody has read the last item's number
    if (0 == threadIdx.x) *slot = atomicAdd(counter, 1u);
    __syncthreads();
    return *slot;
}
// what a workgroup that did list `part` of a cluster leaves behind (status: clusterSums')
__device__ inline void storePart(const SumsBuffers &sb, u32 t, u32 part, u32 status, const ClusterSums &out, u32 *nextList, u32 *nextCount, Counters &local)
{
    if (SUMS_DONE == status)
    {
        if (part < 2) sb.sums[t].shadow[part] = out.shadow[part]; else { sb.sums[t].pair = out.pair; sb.sums[t].ordered = out.ordered; }
        if (0 == part) ++local.largeSums;
    }
    else if (SUMS_TOO_LARGE == status && nextList) { if (0 == part) nextList[atomicAdd(nextCount, 1u)] = t; }       // every part finds the same: the first one says so
    else if (markResidualOnce(sb, t)) { if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else if (SUMS_TOO_LARGE == status) ++local.residualOversize; else ++local.residualCapacity; }
}

// clusterSums (sums.h) for one wavefront and lists of at most 64 entries, arranged for latency instead of generality: lane j
// finishes rescue problem j (the problems' memory round trips overlap), every rescued shadow is fetched once into a table in
// LDS, and the three lists are assembled from that table.  The orders, the near-tie rule and the sums are uniqueSortedSum's.
struct ShadowTable { u64 *pos; double *lp; u32 *obs; u8 *job; };      // SUMS_WAVE_CAP entries each, in LDS

// W lanes of a wavefront per cluster (64, or 16: four clusters per wavefront for the many clusters whose lists are that short); W entries
// per key array and table.  Exchanges stay inside the group: shuffles of width W, votes masked to the group's lanes.
template <u32 W> __device__ inline u32 clusterSumsWave(const DevParams &P, const ClusterFragments &f, const SumInputs &in, SumKeys &k, const ShadowTable &tab, const SumGroup &g, ClusterSums &out, Counters &cnt)
{
    const u32 lane = g.lane, nJobs = in.nJobs;
    if (nJobs > W) { if (W < 64) return SUMS_TOO_LARGE; return clusterSums(P, f, in, k, g, nullptr, true, out, cnt); }
    out.shadow[0] = out.shadow[1] = out.pair = out.ordered = 0.0;
    // the rescue problems, one per lane
    u32 take = 0, side = 0, best = 0, rescued = 0, nGapped = 0, gappedBase = 0, candBase = 0, nCands = 0, retries = 0;
    bool ok = true; double orphanLp = 0.0; ShadowProb orphan; orphan.pos = 0; orphan.logProbability = 0.0; orphan.observedLength = 0;
    if (lane < nJobs)
    {
        RescueJob &job = in.jobs[lane];
        ok = finishRescueFlat(P, job, in, retries);
        take = job.take; side = (job.shadowReadIndex + 1u) % 2; rescued = job.rescued; best = job.rescued ? job.finalBestRank : 0;
        nGapped = job.nGapped; gappedBase = job.gappedBase; candBase = job.candBase; nCands = job.nCands;
        if (take) { const Cand &o = f.list(side)[job.orphanListIndex]; orphan = makeShadowProb(o); orphanLp = o.logProbability; }
    }
    {
        const u32 groupBase = (threadIdx.x & 63u) & ~(W - 1);
        const unsigned long long groupMask = W < 64 ? ((1ull << W) - 1) << groupBase : ~0ull;
        if (__ballot(!ok) & groupMask) return SUMS_RESIDUAL;
    }
    cnt.rescueBsw += retries;
    u32 incl = take;
    for (u32 o = 1; o < W; o <<= 1) { const u32 v = __shfl_up(incl, o, W); if (lane >= o) incl += v; }
    const u32 base = incl - take, total = __shfl(incl, W - 1, W);
    u32 side0 = side ? 0 : take;
    for (int o = W / 2; o > 0; o >>= 1) side0 += __shfl_xor(side0, o, W);
    const u32 nSeeded[2] = { f.nCands[0], f.nCands[1] };
    const u32 shadows[2] = { side0, total - side0 };
    if (shadows[0] + nSeeded[1] > k.cap || shadows[1] + nSeeded[0] > k.cap || total > k.cap) return SUMS_TOO_LARGE;
    // every shadow once: entry base + rank of its problem (the problems of read 1's orphans come first: TemplateBuilder.cpp:737-757).
    // The candidates of all problems are spread over the lanes together -- (problem, candidate) pairs in one sequence -- so that a
    // cluster with several problems pays one memory round trip for their candidate records, not one per problem.
    u32 inclCands = nCands;
    for (u32 o = 1; o < W; o <<= 1) { const u32 v = __shfl_up(inclCands, o, W); if (lane >= o) inclCands += v; }
    const u32 totalCands = __shfl(inclCands, W - 1, W);
    for (u32 idx = lane; idx < ((totalCands + W - 1) & ~(W - 1)); idx += W)
    {
        u32 j = 0;                                             // the problem of candidate idx: the number of problems that end at or before it
        for (u32 q = 0; q < nJobs; ++q) j += (__shfl(inclCands, q, W) <= idx) ? 1u : 0u;
        const u32 jj = j < nJobs ? j : 0;
        const u32 tj = __shfl(take, jj, W), cb = __shfl(candBase, jj, W), bj = __shfl(base, jj, W), endJ = __shfl(inclCands, jj, W), nc = __shfl(nCands, jj, W);
        if (idx >= totalCands || !tj) continue;
        const u32 c = idx - (endJ - nc);
        const Cand &cand = in.shadowCands[cb + c];
        if (!candAligned(cand)) continue;
        const u32 r = in.candRank[cb + c];
        if (r >= tj) continue;
        const ShadowProb p = makeShadowProb(cand);
        tab.pos[bj + r] = p.pos; tab.lp[bj + r] = p.logProbability; tab.obs[bj + r] = u32(p.observedLength); tab.job[bj + r] = u8(jj);
    }
    for (u32 j = 0; j < nJobs; ++j)
    {
        const u32 tj = __shfl(take, j, W);
        const u32 ng = __shfl(nGapped, j, W);
        if (!tj || !__shfl(rescued, j, W) || !ng) continue;
        const u32 bj = __shfl(base, j, W);
        groupSync(g);                                         // the accepted retries replace what the loop above wrote
        const u32 gbase = __shfl(gappedBase, j, W);
        for (u32 kk = lane; kk < ng; kk += W)
        {
            // GappedJob::pad was written a moment ago by the problem's lane: the rule is evaluated again rather than read through memory
            const u32 slot = in.gappedJobs[gbase + kk].tag;
            if (!gappedRetryAccepted(P, in.shadowCands[slot], in.gappedResults[gbase + kk])) continue;
            const u32 r = in.candRank[slot];
            if (r >= tj) continue;
            const ShadowProb p = makeShadowProb(in.gappedResults[gbase + kk].out);
            tab.pos[bj + r] = p.pos; tab.lp[bj + r] = p.logProbability; tab.obs[bj + r] = u32(p.observedLength);
        }
    }
    groupSync(g);
    // sumUniqueShadowProbabilities of either side: its shadows (a contiguous stretch of the table) + the seeded candidates of the other read
    for (u32 s = 0; s < 2; ++s)
    {
        const u32 first = s ? shadows[0] : 0, n = shadows[s] + nSeeded[1 - s];
        if (lane < shadows[s]) { k.pos1[lane] = tab.pos[first + lane]; k.pos2[lane] = 0; k.lp[lane] = tab.lp[first + lane]; k.obs1[lane] = tab.obs[first + lane]; k.obs2[lane] = 0; }
        else if (lane < n) sumKeyFromCand(k, lane, f.list(1 - s)[lane - shadows[s]]);
        groupSync(g);
        if (!uniqueSortedSum(k, n, false, g, nullptr, out.shadow[s])) return SUMS_NEAR_TIE;
    }
    const u32 myJob = lane < total ? tab.job[lane] : 0;
    if (nSeeded[0] && nSeeded[1])
    {   // sumUniquePairProbabilities: every orphan with every shadow it rescued, read 1's alignment first
        const u64 oPos = __shfl(orphan.pos, myJob, W); const double oLp = __shfl(orphan.logProbability, myJob, W);
        const u32 oObs = __shfl(u32(orphan.observedLength), myJob, W), oSide = __shfl(side, myJob, W);
        if (lane < total)
        {
            const u64 sPos = tab.pos[lane]; const double sLp = tab.lp[lane]; const u32 sObs = tab.obs[lane];
            k.pos1[lane] = oSide ? sPos : oPos; k.pos2[lane] = oSide ? oPos : sPos;
            k.lp[lane] = oSide ? sLp + oLp : oLp + sLp;
            k.obs1[lane] = oSide ? sObs : oObs; k.obs2[lane] = oSide ? oObs : sObs;
        }
        groupSync(g);
        if (!uniqueSortedSum(k, total, true, g, nullptr, out.pair)) return SUMS_NEAR_TIE;
    }
    else
    {   // TemplateBuilder::rescueShadow's running sum in list order: the best shadow of a successful rescue has changed places with the first
        const double oLp = __shfl(orphanLp, myJob, W);
        const u32 jBase = __shfl(base, myJob, W), jBest = __shfl(best, myJob, W);
        if (lane < total)
        {
            const u32 r = lane - jBase;
            k.term[jBase + (r == jBest ? 0 : 0 == r ? jBest : r)] = exp(oLp + tab.lp[lane]);
        }
        groupSync(g);
        double sum = 0.0;
        sum = addInOrder(sum, k.term, total);
        out.ordered = sum;
        groupSync(g);
    }
    return SUMS_DONE;
}

// lists of up to 16 entries (most clusters): a quarter of a wavefront per cluster.  What does not fit goes to k_cluster_sums.
#ifndef ISAAC_WAVES_SUMS16
#define ISAAC_WAVES_SUMS16 0
#endif
#if ISAAC_WAVES_SUMS16
__attribute__((amdgpu_waves_per_eu(ISAAC_WAVES_SUMS16, ISAAC_WAVES_SUMS16)))
#endif
__global__ __launch_bounds__(16 * SUMS16_GROUPS) void k_cluster_sums16(DevParams P, ClusterPools pools, u32 nChunk, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters, const u32 *order)
{
    __shared__ __align__(16) u8 keyBytes[SUMS16_GROUPS][SUMS_QUARTER_CAP * 42 + 16];
    __shared__ u64 tabPos[SUMS16_GROUPS][SUMS_QUARTER_CAP]; __shared__ double tabLp[SUMS16_GROUPS][SUMS_QUARTER_CAP]; __shared__ u32 tabObs[SUMS16_GROUPS][SUMS_QUARTER_CAP];
    __shared__ u8 tabJob[SUMS16_GROUPS][SUMS_QUARTER_CAP];
    const u32 group = threadIdx.x >> 4, lane = threadIdx.x & 15;
    const u32 slot = blockIdx.x * SUMS16_GROUPS + group;
    const u32 t = slot < nChunk ? (order ? order[slot] : slot) : nChunk;            // clusters of a kind next to each other (k_cluster_kinds)
    Counters local; memset(&local, 0, sizeof(local));
    if (t < nChunk)
    {
        if (0xffffffffu == rb.jobBase[t]) { if (0 == lane) { markResidual(sb, t); ++local.residualCapacity; } }
        else if (rb.jobCount[t])
        {
            SumKeys keys; sumKeysBind(keys, keyBytes[group], SUMS_QUARTER_CAP);
            SumGroup g; g.lanes = 16; g.lane = lane; g.block = false; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
            ClusterSums out;
            ShadowTable tab; tab.pos = tabPos[group]; tab.lp = tabLp[group]; tab.obs = tabObs[group]; tab.job = tabJob[group];
            const u32 status = clusterSumsWave<16>(P, clusterView(pools.meta[t], pools.cands, pools.cigars), sumInputs(rb, t, gb), keys, tab, g, out, local);
            if (0 == lane)
            {
                if (SUMS_DONE == status) sb.sums[t] = out;
                else if (SUMS_TOO_LARGE == status) sb.mediumList[atomicAdd(sb.mediumCount, 1u)] = t;
                else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else ++local.residualCapacity; }
            }
        }
    }
    flushCounters(local, counters);
}

// lists of up to 64 entries: a wavefront per cluster of the list k_cluster_sums16 left
#ifndef ISAAC_SUMS_WAVES_PER_EU
#define ISAAC_SUMS_WAVES_PER_EU 4      // 128 registers: four wavefronts per SIMD instead of three at 129 (1.96 -> 1.85 ms, profiles/exp_r4_sums_waves.log)
#endif
#if ISAAC_SUMS_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(ISAAC_SUMS_WAVES_PER_EU, ISAAC_SUMS_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(256) void k_cluster_sums(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    __shared__ __align__(16) u8 keyBytes[4][SUMS_WAVE_CAP * 42];
    __shared__ u64 tabPos[4][SUMS_WAVE_CAP]; __shared__ double tabLp[4][SUMS_WAVE_CAP]; __shared__ u32 tabObs[4][SUMS_WAVE_CAP]; __shared__ u8 tabJob[4][SUMS_WAVE_CAP];
    static_assert(SUMS_WAVE_CAP * 42 % 16 == 0, "key arrays stay aligned");
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.mediumCount;
    for (u32 i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4)
    {
        const u32 t = sb.mediumList[i];
        SumKeys keys; sumKeysBind(keys, keyBytes[wave], SUMS_WAVE_CAP);
        SumGroup g; g.lanes = 64; g.lane = lane; g.block = false; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        ClusterSums out;
        ShadowTable tab; tab.pos = tabPos[wave]; tab.lp = tabLp[wave]; tab.obs = tabObs[wave]; tab.job = tabJob[wave];
        const u32 status = clusterSumsWave<64>(P, clusterView(pools.meta[t], pools.cands, pools.cigars), sumInputs(rb, t, gb), keys, tab, g, out, local);
        if (0 == lane)
        {
            if (SUMS_DONE == status) sb.sums[t] = out;
            else if (SUMS_TOO_LARGE == status) sb.midList[atomicAdd(sb.midCount, 1u)] = t;
            else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else ++local.residualCapacity; }
        }
        groupSync(g);
    }
    flushCounters(local, counters);
}

// Lists of up to 256 and of up to 1 024 entries: a workgroup of 256 lanes per list, the keys in LDS -- 10.5 KB for the first kind, which is most of them, so
// that a CU holds twelve such workgroups instead of three.  A cluster whose lists do not fit goes to the next tier's list.
template <u32 CAP> __device__ inline void clusterSumsBlockTier(const DevParams &P, const ClusterPools &pools, const RescueBuffers &rb, const GappedBuffers &gb, const SumsBuffers &sb, Counters *counters,
                                                               const u32 *list, const u32 *listCount, u32 *workCounter, u32 *nextList, u32 *nextCount)
{
    __shared__ __align__(16) u8 keyBytes[CAP * 42];
    __shared__ __align__(16) RescueJob stagedJobs[SUMS_STAGED_JOBS];
    __shared__ u32 scratch, item;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *listCount;
    for (u32 i = nextItem(workCounter, &item); i < 3 * n; i = nextItem(workCounter, &item))       // a workgroup per list, as in the tiers above
    {
        const u32 t = list[i % n], part = 2 - i / n;
        SumKeys keys; sumKeysBind(keys, keyBytes, CAP);
        SumGroup g; g.lanes = 256; g.lane = threadIdx.x; g.block = true; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        ClusterSums out;
        const u32 status = clusterSums(P, clusterView(pools.meta[t], pools.cands, pools.cigars), stagedInputs(rb, t, gb, stagedJobs), keys, g, &scratch, false, out, local, part);
        if (0 == threadIdx.x) storePart(sb, t, part, status, out, nextList, nextCount, local);
        __syncthreads();
    }
    flushCounters(local, counters);
}
__global__ __launch_bounds__(256) void k_cluster_sums_mid(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{ clusterSumsBlockTier<SUMS_MID_CAP>(P, pools, rb, gb, sb, counters, sb.midList, sb.midCount, sb.midCount + 1, sb.largeList, sb.largeCount); }
__global__ __launch_bounds__(256) void k_cluster_sums_large(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{ clusterSumsBlockTier<SUMS_BLOCK_CAP>(P, pools, rb, gb, sb, counters, sb.largeList, sb.largeCount, sb.largeCount + 6, sb.xlList, sb.xlCount); }

// lists of up to 3584 entries: 1024 lanes per cluster, the keys in 147 KB of the CU's 160 KB of LDS
__global__ __launch_bounds__(1024) void k_cluster_sums_xl(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    extern __shared__ __align__(16) u8 xlKeyBytes[];
    __shared__ u32 scratch, item;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.xlCount;
    SumKeys keys; sumKeysBind(keys, xlKeyBytes, SUMS_XL_CAP);
    for (u32 i = nextItem(sb.xlCount + 3, &item); i < 3 * n; i = nextItem(sb.xlCount + 3, &item))       // a workgroup per list: (cluster, part)
    {
        const u32 t = sb.xlList[i % n], part = 2 - i / n;
        SumGroup g; g.lanes = 1024; g.lane = threadIdx.x; g.block = true; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        ClusterSums out;
        const u32 status = clusterSums(P, clusterView(pools.meta[t], pools.cands, pools.cigars), sumInputs(rb, t, gb), keys, g, &scratch, false, out, local, part);
        if (0 == threadIdx.x) storePart(sb, t, part, status, out, sb.hugeList, sb.hugeCount, local);
        __syncthreads();
    }
    flushCounters(local, counters);
}

// the clusters of repeat families: lists of thousands of entries (up to the reference's own 32768), keys in HBM (L2-resident: 1.4 MB
// per workgroup), 1024 lanes per cluster
__global__ __launch_bounds__(1024) void k_cluster_sums_huge(DevParams P, ClusterPools pools, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    __shared__ u32 scratch, item;
    __shared__ __align__(16) u16 radixCounts[16 * 1024];
    __shared__ u32 radixTotals[1024];
    __shared__ u64 radixVary[2];
    __shared__ u8 radixDigits[SUMS_HUGE_DIGITS];
    __shared__ u16 closeIdx[2 * SUMS_HUGE_CLOSE_IDX];
    __shared__ __align__(16) RescueJob stagedJobs[SUMS_STAGED_JOBS];
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.hugeCount;
    u8 *mine = sb.hugeKeys + size_t(blockIdx.x) * SUMS_HUGE_CAP * SUMS_HUGE_ENTRY;
    SumKeys keys; sumKeysBind(keys, mine, SUMS_HUGE_CAP);
    for (u32 i = nextItem(sb.hugeCount + 3, &item); i < 3 * n; i = nextItem(sb.hugeCount + 3, &item))       // a workgroup per list: (cluster, part)
    {
        const u32 t = sb.hugeList[i % n], part = 2 - i / n;
        SumGroup g; g.lanes = 1024; g.lane = threadIdx.x; g.block = true; g.radixMin = 0;
        g.sumTile = reinterpret_cast<double *>(radixCounts); g.sumTileCap = sizeof(radixCounts) / 8;        // the counts are idle by then
        g.radix.counts = radixCounts; g.radix.totals = radixTotals; g.radix.vary = radixVary; g.radix.alt = reinterpret_cast<u16 *>(mine + size_t(SUMS_HUGE_CAP) * 42); g.radix.digits = radixDigits; g.radix.digitsCap = SUMS_HUGE_DIGITS;
        g.radix.closeIdx = closeIdx; g.radix.closeIdxCap = SUMS_HUGE_CLOSE_IDX;
        ClusterSums out;
        const u32 status = clusterSums(P, clusterView(pools.meta[t], pools.cands, pools.cigars), stagedInputs(rb, t, gb, stagedJobs), keys, g, &scratch, false, out, local, part);
        if (0 == threadIdx.x) storePart(sb, t, part, status, out, nullptr, nullptr, local);
        __syncthreads();
    }
    flushCounters(local, counters);
}
