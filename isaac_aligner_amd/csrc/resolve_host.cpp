// One cluster through the thread-serial form of the path on the host: FragmentBuilder::build, TemplateBuilder::buildTemplate with its mate rescues, the end
// clippers and the FragmentHeader records, from the cluster's seed matches -- the very headers the kernels are made of (csrc/*.h), compiled by g++, with
// glibc's exp / log10 / floor where the device has its own maths library.  What isaac_gpu_resolve_flagged runs for the few clusters per million whose
// MAPQ arithmetic came within 1e-11 of an integer on the device (isaac_fragment::reserved bit 3): there, and only there, a last-ulp difference between
// the two libraries could move a floor(-10 log10 x) (lib/alignment/TemplateBuilder.cpp:270-273,433-439), so those clusters take glibc's answer, which is
// the reference's.  Not a CPU path of the product: nothing else is ever computed here.
#include "cluster_ops.h"
#include "host_util.h"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace isaac;

namespace isaac_host_resolve
{

struct Resolver
{
    DevParams P; DevReference R;
    DevAdapters adapters; u32 adapterRanges[4];
    std::vector<u64> contigOffset; std::vector<u8> contigLoaded;      // copies: the context's vectors may be reassigned while this lives
    std::vector<double> logMatch, logMismatch;
    RogCorrection rog; double lmq40;
    std::vector<u8> arena; TemplateWork work;
    FragmentWorkStore fragmentStore; FragmentWork fragmentWork;
    ClusterStore store;
};

Resolver *create(const isaac_params &params, const char *bases, const u64 *contigOffset, const u8 *contigLoaded, u32 nContigs)
{
    std::unique_ptr<Resolver> r(new Resolver);
    r->P = makeDevParams(params);
    r->logMatch.resize(100); r->logMismatch.resize(100); makeQualityTables(r->logMatch.data(), r->logMismatch.data());
    std::memset(&r->R, 0, sizeof(r->R));
    r->contigOffset.assign(contigOffset, contigOffset + nContigs + 1); r->contigLoaded.assign(contigLoaded, contigLoaded + nContigs);
    contigOffset = r->contigOffset.data(); contigLoaded = r->contigLoaded.data();
    r->R.bases = bases; r->R.totalBases = contigOffset[nContigs]; r->R.contigOffset = contigOffset; r->R.contigLoaded = contigLoaded; r->R.nContigs = nContigs;
    r->R.logMatch = r->logMatch.data(); r->R.logMismatch = r->logMismatch.data(); r->R.logStride = 1;
    r->rog = makeRogCorrection(r->P, contigOffset, contigLoaded, nContigs);
    r->lmq40 = logMismatchQ40();
    const TemplateCaps caps = heavyCaps();                      // the reference's own capacities
    r->arena.assign(templateWorkBytes(caps) + 16, 0);
    templateWorkBind(r->work, reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(r->arena.data()) + 15) & ~uintptr_t(15)), caps);
    r->fragmentWork = r->fragmentStore.bind();
    r->adapters = makeDevAdapters(params);
    if (r->adapters.n) { r->P.adapters = &r->adapters; r->P.adapterRanges = r->adapterRanges; r->P.adapterCandBase = r->store.cands; }     // (one cluster at a time: its view starts at the store's first slot)
    return r.release();
}
void destroy(Resolver *r) { delete r; }

// clusterBcl: the cluster's BCL bytes; matches: its seed matches as isaac_gpu_find_matches left them; records: n_reads records, cigars: n_reads x 40 words (the
// records' cigar_offset is relative to `cigars`)
void selectCluster(Resolver *r, const isaac_tls &tls, const u8 *clusterBcl, u32 cluster, u32 tile, const Match *matches, u32 nMatches, FragmentRecord *records, u32 *cigars)
{
    DevTls t; std::memcpy(&t, &tls, sizeof(t));
    Counters cnt; std::memset(&cnt, 0, sizeof(cnt));
    // the per-cluster functions index the tile by cluster: a tile of one, whose only cluster has the number the records must carry
    ClusterFragments f = r->store.view();
    const u64 offsets[2] = { 0, nMatches };
    // (clusterBuildFragments adds cluster * clusterLength to the BCL pointer and reads offsets[cluster]: both are given for cluster 0)
    clusterBuildFragments(r->P, r->R, clusterBcl, 0, matches, offsets, true, true, r->fragmentWork, f, cnt, false);
    clusterFinishFragments(r->P, r->R, clusterBcl, 0, true, nullptr, r->fragmentWork, f, cnt);
    CoopInputs coop; coop.lanes = 1; coop.lane = 0; coop.fastSort = false; coop.ldsSort = nullptr; coop.ldsSortCap = 0;
    // records and CIGARs of cluster `cluster` land at index cluster * n_reads: offset the outputs so that they land at the caller's
    FragmentRecord *recordBase = records - u64(cluster) * r->P.nReads;
    u32 *cigarBase = cigars - u64(cluster) * r->P.nReads * OUT_CIGAR_CAP;
    clusterSelect(r->P, r->R, t, r->rog, r->lmq40, clusterBcl - u64(cluster) * r->P.clusterLength, cluster, tile, f, r->work, recordBase, cigarBase, cnt, nullptr, &coop);
}

} // namespace isaac_host_resolve
