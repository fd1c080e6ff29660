// Per-cluster entry points of the thread-serial stages: one GPU thread (or one loop iteration of tests/hostemu) runs one of
// these for one cluster.  MatchSelector::processMatchList / determineTemplateLength (lib/alignment/MatchSelector.cpp:188-368).
#pragma once
#include "template.h"

namespace isaac
{

static const u32 OUT_CIGAR_CAP = 40;   // ISAAC_GPU_MAX_CIGAR_OPS words per read in the select output
// isaac_fragment::reserved
enum { RECORD_TEMPLATE_OVERFLOW = 1,   // a template-stage work list overflowed: redo the cluster with heavyCaps()
       RECORD_NOT_STORED = 2,          // the reference would not have stored this template (only without --keep-unaligned)
       RECORD_FRAGMENT_OVERFLOW = 4 }; // a fragment-stage capacity was exceeded: the cluster's result is not exact

// FragmentBuilder::build for cluster `cluster` of the tile; its matches are matches[offsets[cluster] .. offsets[cluster + 1])
ISAAC_HD void clusterBuildFragments(const DevParams &P, const DevReference &R, const u8 *bcl, u32 cluster, const Match *matches, const u64 *offsets,
                                    bool withGaps, bool trim, FragmentWork &work, ClusterFragments &out, Counters &cnt)
{
    const u64 begin = offsets[cluster], end = offsets[cluster + 1];
    buildFragments(P, R, bcl + u64(cluster) * P.clusterLength, matches + begin, u32(end - begin), withGaps, trim, work, out, cnt);
    if (out.flags & CLUSTER_OVERFLOW) ++cnt.overflowClusters;
}

// what TemplateLengthDistribution::addTemplate needs to know about the cluster (TemplateLengthStatistics.cpp:275-314)
ISAAC_HD void clusterTlsSample(const ClusterFragments &f, u32 nMatches, TlsSample &s)
{
    s.valid = u8(nMatches != 0 && f.built); // a cluster whose match list starts with NoMatch is skipped (MatchSelector.cpp:230)
    s.n0 = f.nCands[0]; s.n1 = f.nCands[1];
    s.contig0 = s.contig1 = 0; s.pos0 = s.pos1 = 0; s.obs0 = s.obs1 = 0; s.rev0 = s.rev1 = 0; s.insertEnd = 0;
    if (1 == s.n0 && 1 == s.n1)
    {
        const Cand &a = f.cands[0][0], &b = f.cands[1][0];
        s.contig0 = a.contigId; s.contig1 = b.contigId; s.pos0 = a.position; s.pos1 = b.position; s.obs0 = a.observedLength; s.obs1 = b.observedLength;
        s.rev0 = a.reverse; s.rev1 = b.reverse;
        const u32 a0 = f.cigarPool[a.cigarOffset] & 0xf, a1 = f.cigarPool[a.cigarOffset + a.cigarLength - 1] & 0xf;
        const u32 b0 = f.cigarPool[b.cigarOffset] & 0xf, b1 = f.cigarPool[b.cigarOffset + b.cigarLength - 1] & 0xf;
        s.insertEnd = u8(a0 == OP_INSERT || a1 == OP_INSERT || b0 == OP_INSERT || b1 == OP_INSERT);
    }
}

// MatchSelector::processMatchList for one cluster: template building, clipping, io::FragmentHeader records.
// records: P.nReads per cluster; cigars: P.nReads * OUT_CIGAR_CAP words per cluster
ISAAC_HD void clusterSelect(const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, double logMismatchQ40,
                            const u8 *bcl, u32 cluster, u32 tile, const ClusterFragments &frags, TemplateWork &work,
                            FragmentRecord *records, u32 *cigars, Counters &cnt)
{
    TemplateCtx x;
    x.P = &P; x.R = &R; x.tls = &tls; x.frags = &frags; x.w = &work; x.cnt = &cnt; x.clusterId = cluster;
    x.rogRead[0] = rog.read[0]; x.rogRead[1] = rog.read[1]; x.rog = rog.pair;
    const u8 *clusterBcl = bcl + u64(cluster) * P.clusterLength;
    for (u32 r = 0; r < 2; ++r)
    {
        x.reads[r].bcl = clusterBcl + P.readOffset[r]; x.reads[r].length = r < P.nReads ? P.readLength[r] : 0;
        x.reads[r].firstCycle = P.firstCycle[r]; x.reads[r].endCyclesMasked = frags.endCyclesMasked[r];
    }
    BamTemplate t;
    const bool store = selectCluster(x, t, logMismatchQ40);
    for (u32 i = 0; i < P.nReads; ++i)
    {
        FragmentRecord &r = records[u64(cluster) * P.nReads + i];
        u32 *cig = cigars + (u64(cluster) * P.nReads + i) * OUT_CIGAR_CAP;
        if (!store) { bamTemplateInitialize(x, t); }
        makeFragmentRecord(x, t, i, tile, r);
        r.cigarOffset = u32((u64(cluster) * P.nReads + i) * OUT_CIGAR_CAP);
        const Cand &f = t.f[i].c;
        u32 n = f.cigarLength;
        if (n > OUT_CIGAR_CAP) { n = OUT_CIGAR_CAP; r.reserved |= RECORD_FRAGMENT_OVERFLOW; }
        for (u32 k = 0; k < n; ++k) cig[k] = t.f[i].pool[f.cigarOffset + k];
        if (work.overflow) r.reserved |= RECORD_TEMPLATE_OVERFLOW;
        if (frags.flags & CLUSTER_OVERFLOW) r.reserved |= RECORD_FRAGMENT_OVERFLOW;
        if (!store) r.reserved |= RECORD_NOT_STORED;
    }
}

} // namespace isaac
