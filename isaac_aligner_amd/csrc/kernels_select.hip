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
    // A thread's copy of its cluster's candidate lists and of its rescue problems' outcomes (template_lean.h walks the lists several times, every
    // walk a chain of dependent loads; clusters of a kind lie anywhere in the pool, so 64 lanes touch 64 unrelated lines each time and the caches
    // hold none of them until the next walk).  All of it is asked for at once -- one round trip -- and read from LDS afterwards.  A thread's area is
    // an odd number of 8-byte words, so that the lanes' reads of one field of one list entry spread over the banks.
#if ISAAC_SELECT_STAGE
    __shared__ u64 stage[SELECT_BLOCK * SELECT_STAGE_WORDS];
#endif
    const DevParams &P = constants->P; const DevTls &tls = constants->tls; const RogCorrection &rog = constants->rog;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = slot < nChunk ? (order ? order[slot] : slot) : nChunk;        // clusters of a kind next to each other: see k_cluster_kinds
    u32 mapqNearInteger = 0;
    if (t < nChunk && !skip[t])
    {
        const ClusterMeta meta = pools.meta[t];
        LeanRescue rs;
        rs.jobs = rb.jobs + rb.jobBase[t]; rs.jobCount = rb.jobCount[t]; rs.shadowCands = rb.shadowCands; rs.shadowCigars = rb.shadowCigars; rs.gappedResults = gappedResults; rs.sums = sums + t;
        const Cand *l0 = pools.cands + meta.first, *l1 = l0 + meta.second;
        const u32 n0 = meta.nCands[0], n1 = meta.nCands[1];
        bool done;
#if ISAAC_SELECT_STAGE
        if (meta.built && n0 + n1 <= SELECT_STAGE && rs.jobCount <= SELECT_STAGE)
        {
            Cand *mine = reinterpret_cast<Cand *>(stage + threadIdx.x * SELECT_STAGE_WORDS);
            RescueOutcome *outcomes = reinterpret_cast<RescueOutcome *>(mine + SELECT_STAGE);
            // every load first, then the stores: nothing below waits for one entry before asking for the next
            uint4 v[SELECT_STAGE][4]; uint2 o[SELECT_STAGE][3];
#pragma unroll
            for (u32 i = 0; i < SELECT_STAGE; ++i)
                if (i < n0 + n1)
                {
                    const uint4 *from = reinterpret_cast<const uint4 *>(i < n0 ? l0 + i : l1 + (i - n0));
                    v[i][0] = from[0]; v[i][1] = from[1]; v[i][2] = from[2]; v[i][3] = from[3];
                }
#pragma unroll
            for (u32 k = 0; k < SELECT_STAGE; ++k)
                if (k < rs.jobCount)
                {
                    const uint2 *from = reinterpret_cast<const uint2 *>(&rs.jobs[k].out);
                    o[k][0] = from[0]; o[k][1] = from[1]; o[k][2] = from[2];
                }
#pragma unroll
            for (u32 i = 0; i < SELECT_STAGE; ++i)
                if (i < n0 + n1)
                {
                    uint2 *to = reinterpret_cast<uint2 *>(mine + i);
                    to[0] = make_uint2(v[i][0].x, v[i][0].y); to[1] = make_uint2(v[i][0].z, v[i][0].w); to[2] = make_uint2(v[i][1].x, v[i][1].y); to[3] = make_uint2(v[i][1].z, v[i][1].w);
                    to[4] = make_uint2(v[i][2].x, v[i][2].y); to[5] = make_uint2(v[i][2].z, v[i][2].w); to[6] = make_uint2(v[i][3].x, v[i][3].y); to[7] = make_uint2(v[i][3].z, v[i][3].w);
                }
#pragma unroll
            for (u32 k = 0; k < SELECT_STAGE; ++k)
                if (k < rs.jobCount)
                {
                    uint2 *to = reinterpret_cast<uint2 *>(outcomes + k);
                    to[0] = o[k][0]; to[1] = o[k][1]; to[2] = o[k][2];
                }
            rs.outcomes = reinterpret_cast<const u8 *>(outcomes); rs.outcomeStride = u32(sizeof(RescueOutcome));
            done = leanSelectCluster(P, R, tls, rog, logMismatchQ40, bcl, clusterBase + t, tile, meta, mine, mine + n0, pools.cigars, rs, records, cigars, mapqNearInteger);
        }
        else
#endif
        {
            leanOutcomesInPlace(rs);
            done = leanSelectCluster(P, R, tls, rog, logMismatchQ40, bcl, clusterBase + t, tile, meta, l0, l1, pools.cigars, rs, records, cigars, mapqNearInteger);
        }
#if ISAAC_TINY_BEST < 4
        if (t & 1) done = false;      // the test build of tests/test_gpu_parity.py::test_residual_pass_on_most_clusters: every other cluster takes the residual pass
#endif
        if (!done) { mapqNearInteger = 0; overflowList[atomicAdd(overflowCount, 1u)] = t; }
    }
    flushCounter(&Counters::clusters, t < nChunk ? 1u : 0u, counters);   // including the ones the wave-per-cluster pass takes
    flushCounter(&Counters::mapqNearInteger, mapqNearInteger, counters);
}
