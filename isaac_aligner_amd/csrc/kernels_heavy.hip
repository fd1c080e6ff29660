// kernels_heavy.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"

// k_select_heavy: one wave per cluster of the overflow list (sums == NULL: the clusters the sums kernels could not finish; else the
// clusters whose placements overflowed k_select's private lists).  All 64 lanes execute the template logic together on one arena
// (same statements, same data), which costs what one thread costs; the bulk steps (probability sorts) are spread over the lanes.
__global__ __launch_bounds__(64) void k_select_heavy(DevParams P, DevReference R, DevTls tls, RogCorrection rog, double logMismatchQ40, const u8 *bcl, u32 clusterBase, u32 nList, const u32 *nListDev, u32 tile,
                                                     ClusterPools pools, u8 *arena, u64 arenaBytes, TemplateCaps caps, const u32 *list, RescueBuffers rb, const GappedResult *gappedResults, const GappedJob *gappedJobs,
                                                     const ClusterSums *sums, FragmentRecord *records, u32 *cigars, Counters *counters)
{
    extern __shared__ __align__(16) u8 heavyLds[];
    Counters local; memset(&local, 0, sizeof(local));
    const u32 n = nListDev ? *nListDev : nList;
    for (u32 t = blockIdx.x; t < n; t += gridDim.x)
    {
    const u32 inChunk = list[t];
    TemplateWork work;
    templateWorkBind(work, arena + u64(blockIdx.x) * arenaBytes, caps);
    RescueInputs in; const RescueInputs *pin = nullptr;
    if (rb.jobBase && rb.jobBase[inChunk] != 0xffffffffu)
    {
        in.jobs = rb.jobs + rb.jobBase[inChunk]; in.jobCount = rb.jobCount[inChunk]; in.shadowCands = rb.shadowCands; in.shadowCigars = rb.shadowCigars; in.candRank = rb.candRank;
        in.gappedResults = gappedResults; in.gappedJobs = gappedJobs; in.serialFallbackAllowed = true;
        // sums: the cluster's rescue outcomes and probability sums are there already (it is here because k_select's small lists overflowed)
        in.sums = sums ? sums + inChunk : nullptr;
        pin = &in;
    }
    CoopInputs coop; coop.lanes = 64; coop.lane = threadIdx.x; coop.fastSort = true; coop.ldsSort = reinterpret_cast<u16 *>(heavyLds); coop.ldsSortCap = HEAVY_SORT_LDS;
    clusterSelect(P, R, tls, rog, logMismatchQ40, bcl, clusterBase + inChunk, tile, clusterView(pools.meta[inChunk], pools.cands, pools.cigars), work, records, cigars, local, pin, &coop);
    if (work.overflow) ++local.overflowClusters;   // even the reference's own capacities were exceeded
    ++local.heavyClusters;
    __syncthreads();                                 // the arena is reused by the block's next cluster
    }
    if (0 != threadIdx.x) memset(&local, 0, sizeof(local));   // every lane counted the same events
    flushCounters(local, counters);
}

