// kernels_sums.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"

__device__ inline SumInputs sumInputs(const RescueBuffers &rb, u32 t, const GappedBuffers &gb)
{
    SumInputs in; in.jobs = rb.jobs + rb.jobBase[t]; in.nJobs = rb.jobCount[t]; in.shadowCands = rb.shadowCands; in.candRank = rb.candRank; in.gappedResults = gb.results; in.gappedJobs = gb.jobs;
    return in;
}
__device__ inline void markResidual(const SumsBuffers &sb, u32 t) { sb.residualFlag[t] = 1; sb.residualList[atomicAdd(sb.residualCount, 1u)] = t; }

__global__ __launch_bounds__(256) void k_cluster_sums(DevParams P, const ClusterFragments *frags, u32 nChunk, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    __shared__ __align__(16) u8 keyBytes[4][SUMS_WAVE_CAP * 42];
    static_assert(SUMS_WAVE_CAP * 42 % 16 == 0, "key arrays stay aligned");
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 t = blockIdx.x * 4 + wave;
    Counters local; memset(&local, 0, sizeof(local));
    if (t < nChunk)
    {
        if (0xffffffffu == rb.jobBase[t]) { if (0 == lane) { markResidual(sb, t); ++local.residualCapacity; } }
        else if (rb.jobCount[t])
        {
            SumKeys keys; sumKeysBind(keys, keyBytes[wave], SUMS_WAVE_CAP);
            SumGroup g; g.lanes = 64; g.lane = lane; g.block = false; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
            ClusterSums out;
            const u32 status = clusterSums(P, frags[t], sumInputs(rb, t, gb), keys, g, nullptr, true, out, local);
            if (0 == lane)
            {
                if (SUMS_DONE == status) sb.sums[t] = out;
                else if (SUMS_TOO_LARGE == status) sb.largeList[atomicAdd(sb.largeCount, 1u)] = t;
                else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else ++local.residualCapacity; }
            }
        }
    }
    flushCounters(local, counters);
}

__global__ __launch_bounds__(256) void k_cluster_sums_large(DevParams P, const ClusterFragments *frags, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    __shared__ __align__(16) u8 keyBytes[SUMS_BLOCK_CAP * 42];
    __shared__ u32 scratch;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.largeCount;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x)
    {
        const u32 t = sb.largeList[i];
        SumKeys keys; sumKeysBind(keys, keyBytes, SUMS_BLOCK_CAP);
        SumGroup g; g.lanes = 256; g.lane = threadIdx.x; g.block = true; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        ClusterSums out;
        const u32 status = clusterSums(P, frags[t], sumInputs(rb, t, gb), keys, g, &scratch, false, out, local);
        if (0 == threadIdx.x)
        {
            if (SUMS_DONE == status) { sb.sums[t] = out; ++local.largeSums; }
            else if (SUMS_TOO_LARGE == status) sb.xlList[atomicAdd(sb.xlCount, 1u)] = t;
            else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else ++local.residualCapacity; }
        }
        __syncthreads();
    }
    flushCounters(local, counters);
}

// lists of up to 3584 entries: 1024 lanes per cluster, the keys in 147 KB of the CU's 160 KB of LDS
__global__ __launch_bounds__(1024) void k_cluster_sums_xl(DevParams P, const ClusterFragments *frags, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    extern __shared__ __align__(16) u8 xlKeyBytes[];
    __shared__ u32 scratch;
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.xlCount;
    SumKeys keys; sumKeysBind(keys, xlKeyBytes, SUMS_XL_CAP);
    for (u32 i = blockIdx.x; i < n; i += gridDim.x)
    {
        const u32 t = sb.xlList[i];
        SumGroup g; g.lanes = 1024; g.lane = threadIdx.x; g.block = true; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        ClusterSums out;
        const u32 status = clusterSums(P, frags[t], sumInputs(rb, t, gb), keys, g, &scratch, false, out, local);
        if (0 == threadIdx.x)
        {
            if (SUMS_DONE == status) { sb.sums[t] = out; ++local.largeSums; }
            else if (SUMS_TOO_LARGE == status) sb.hugeList[atomicAdd(sb.hugeCount, 1u)] = t;
            else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else ++local.residualCapacity; }
        }
        __syncthreads();
    }
    flushCounters(local, counters);
}

// the clusters of repeat families: lists of thousands of entries (up to the reference's own 32768), keys in HBM (L2-resident: 1.4 MB
// per workgroup), 1024 lanes per cluster
__global__ __launch_bounds__(1024) void k_cluster_sums_huge(DevParams P, const ClusterFragments *frags, RescueBuffers rb, GappedBuffers gb, SumsBuffers sb, Counters *counters)
{
    __shared__ u32 scratch;
    __shared__ __align__(16) u16 radixCounts[16 * 1024];
    __shared__ u32 radixTotals[1024];
    __shared__ u64 radixVary[2];
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = *sb.hugeCount;
    u8 *mine = sb.hugeKeys + size_t(blockIdx.x) * SUMS_HUGE_CAP * SUMS_HUGE_ENTRY;
    SumKeys keys; sumKeysBind(keys, mine, SUMS_HUGE_CAP);
    for (u32 i = blockIdx.x; i < n; i += gridDim.x)
    {
        const u32 t = sb.hugeList[i];
        SumGroup g; g.lanes = 1024; g.lane = threadIdx.x; g.block = true; g.radixMin = 0;
        g.sumTile = reinterpret_cast<double *>(radixCounts); g.sumTileCap = sizeof(radixCounts) / 8;        // the counts are idle by then
        g.radix.counts = radixCounts; g.radix.totals = radixTotals; g.radix.vary = radixVary; g.radix.alt = reinterpret_cast<u16 *>(mine + size_t(SUMS_HUGE_CAP) * 42);
        ClusterSums out;
        const u32 status = clusterSums(P, frags[t], sumInputs(rb, t, gb), keys, g, &scratch, false, out, local);
        if (0 == threadIdx.x)
        {
            if (SUMS_DONE == status) { sb.sums[t] = out; ++local.largeSums; }
            else { markResidual(sb, t); if (SUMS_NEAR_TIE == status) ++local.residualNearTie; else if (SUMS_TOO_LARGE == status) ++local.residualOversize; else ++local.residualCapacity; }
        }
        __syncthreads();
    }
    flushCounters(local, counters);
}

