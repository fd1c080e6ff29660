// kernels_select.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"

// k_select: TemplateBuilder::buildTemplate on the precomputed rescue outcomes and sums, the clippers and the FragmentHeader records,
// one thread per cluster of the chunk; `skip`: clusters the wave-per-cluster pass takes.  Clusters whose private work area overflows
// (more equally good placements than it holds) are appended to overflowList and redone by that pass as well.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ISAAC_SELECT_WAVES))) void k_select(const TemplateConstants *constants, DevReference R, double logMismatchQ40, const u8 *bcl, u32 clusterBase, u32 nChunk, u32 tile,
                                               ClusterPools pools, RescueBuffers rb, const GappedResult *gappedResults, const GappedJob *gappedJobs, const ClusterSums *sums,
                                               FragmentRecord *records, u32 *cigars, u32 *overflowList, u32 *overflowCount, const u8 *skip, Counters *counters, const u32 *order)
{
    const DevParams &P = constants->P; const DevTls &tls = constants->tls; const RogCorrection &rog = constants->rog;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = slot < nChunk ? (order ? order[slot] : slot) : nChunk;        // clusters of a kind next to each other: see k_cluster_kinds
    Counters local; memset(&local, 0, sizeof(local));
    if (t < nChunk && !skip[t])
    {
        __attribute__((aligned(16))) u8 workBytes[TINY_WORK_BYTES];
        TemplateWork work;
        templateWorkBind(work, workBytes, tinyCaps());
        RescueInputs in;
        in.jobs = rb.jobs + rb.jobBase[t]; in.jobCount = rb.jobCount[t]; in.shadowCands = rb.shadowCands; in.shadowCigars = rb.shadowCigars; in.candRank = rb.candRank;
        in.gappedResults = gappedResults; in.gappedJobs = gappedJobs; in.serialFallbackAllowed = false; in.sums = sums + t;
        Cand privateCands[2 * PRIVATE_CANDS];
        clusterSelect(P, R, tls, rog, logMismatchQ40, bcl, clusterBase + t, tile, clusterView(pools.meta[t], pools.cands, pools.cigars), work, records, cigars, local, &in, nullptr, privateCands);
        if (work.overflow) overflowList[atomicAdd(overflowCount, 1u)] = t;
    }
    if (t < nChunk) ++local.clusters;   // including the ones the wave-per-cluster pass takes
    flushCounters(local, counters);
}

