// Host side of the FASTQ input (no GPU work): how a lane's clusters become tiles.  FastqSeedSource (lib/workflow/alignWorkflow/
// FastqDataSource.cpp:82-84 tileClustersMax_, :103-176 discoverTiles): a load of up to --clusters-at-a-time clusters is cut into tiles of at
// most 40 000 000 / #seeds clusters, numbered on from the lane's previous load; cluster ids restart in every tile (they are the SeedId's 31-bit
// cluster field and the number in the BAM read name).  Without --clusters-at-a-time the reference sizes a load by how much memory it can
// allocate (determineMemoryCapacity), which makes tile boundaries depend on the machine: parity runs pass the option.
#include "../../include/isaac_gpu.h"

#include <algorithm>

extern "C" {

uint32_t isaac_gpu_fastq_tile_clusters_max(uint32_t clustersAtATime, uint32_t nSeeds)
{
    if (!nSeeds) return 0;
    const uint32_t bySeeds = 40000000u / nSeeds;
    return clustersAtATime ? std::min(clustersAtATime, bySeeds) : bySeeds;
}

int isaac_gpu_fastq_tiles(uint32_t clustersLoaded, uint32_t clustersAtATime, uint32_t nSeeds, uint32_t firstTile,
                          uint32_t *tileNumbers, uint32_t *tileClusters, uint32_t capacity, uint32_t *nTilesOut, uint32_t *nextTileOut)
{
    if (nTilesOut) *nTilesOut = 0;
    if (nextTileOut) *nextTileOut = firstTile;
    const uint32_t tileClustersMax = isaac_gpu_fastq_tile_clusters_max(clustersAtATime, nSeeds);
    if (!tileClustersMax) return ISAAC_GPU_EINVAL;
    uint32_t n = 0, tile = firstTile;
    while (clustersLoaded)
    {
        const uint32_t clusterCount = std::min(clustersLoaded, tileClustersMax);
        if (n < capacity) { if (tileNumbers) tileNumbers[n] = tile; if (tileClusters) tileClusters[n] = clusterCount; }
        ++n; ++tile;
        if (clustersLoaded < tileClustersMax) break;
        clustersLoaded -= tileClustersMax;
    }
    if (nTilesOut) *nTilesOut = n;
    if (nextTileOut) *nextTileOut = tile;
    return n > capacity ? ISAAC_GPU_ECAPACITY : 0;
}

} // extern "C"
