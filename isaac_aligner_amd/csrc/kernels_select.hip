// kernels_select.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"
#include "template_lean.h"

// k_select: TemplateBuilder::buildTemplate on the precomputed rescue outcomes and sums, the clippers and the FragmentHeader records,
// one thread per cluster of the chunk (template_lean.h: no work area, no scratch); `skip`: clusters the wave-per-cluster pass takes.
// Clusters the lean form does not do (leanSelectCluster) are appended to overflowList and redone by that pass as well.
#ifndef ISAAC_WAVES_SELECT
#define ISAAC_WAVES_SELECT 0
#endif
#if ISAAC_WAVES_SELECT
__attribute__((amdgpu_waves_per_eu(ISAAC_WAVES_SELECT, ISAAC_WAVES_SELECT)))
#endif
__global__ __launch_bounds__(SELECT_BLOCK) void k_select(const TemplateConstants *__restrict__ constants, DevReference R, double logMismatchQ40, const u8 *__restrict__ bcl, u32 clusterBase, u32 nChunk, u32 tile,
                                               ClusterPools pools, RescueBuffers rb, const GappedResult *__restrict__ gappedResults, const ClusterSums *__restrict__ sums,
                                               FragmentRecord *__restrict__ records, u32 *__restrict__ cigars, u32 *overflowList, u32 *overflowCount, const u8 *__restrict__ skip, Counters *counters, const u32 *__restrict__ order)
{
    const DevParams &P = constants->P; const DevTls &tls = constants->tls; const RogCorrection &rog = constants->rog;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = slot < nChunk ? (order ? order[slot] : slot) : nChunk;        // clusters of a kind next to each other: see k_cluster_kinds
    u32 mapqNearInteger = 0;
    if (t < nChunk && !skip[t])
    {
        LeanRescue rs;
        rs.jobs = rb.jobs + rb.jobBase[t]; rs.jobCount = rb.jobCount[t]; rs.shadowCands = rb.shadowCands; rs.shadowCigars = rb.shadowCigars; rs.gappedResults = gappedResults; rs.sums = sums + t;
        bool done = leanSelectCluster(P, R, tls, rog, logMismatchQ40, bcl, clusterBase + t, tile, pools.meta[t], pools.cands, pools.cigars, rs, records, cigars, mapqNearInteger);
#if ISAAC_TINY_BEST < 4
        if (t & 1) done = false;      // the test build of tests/test_gpu_parity.py::test_residual_pass_on_most_clusters: every other cluster takes the residual pass
#endif
        if (!done) { mapqNearInteger = 0; overflowList[atomicAdd(overflowCount, 1u)] = t; }
    }
    flushCounter(&Counters::clusters, t < nChunk ? 1u : 0u, counters);   // including the ones the wave-per-cluster pass takes
    flushCounter(&Counters::mapqNearInteger, mapqNearInteger, counters);
}
