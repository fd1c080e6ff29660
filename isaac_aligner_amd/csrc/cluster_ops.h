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
       RECORD_FRAGMENT_OVERFLOW = 4,   // a fragment-stage capacity was exceeded: the cluster's result is not exact
       RECORD_MAPQ_NEAR_INTEGER = 8,
       RECORD_CIGAR_REALIGNED = 16 };  // BAM stage, on its private copy of the records: the CIGAR lies in the gap realigner's pool (BamTile::cigarsAlt) // a MAPQ of this cluster is floor(v) with v within 1e-11 of an integer (mapqFloor): a host that must be sure
                                       // the device's log10/exp rounded like glibc's re-derives exactly these clusters

// FragmentBuilder::build for cluster `cluster` of the tile; its matches are matches[offsets[cluster] .. offsets[cluster + 1])
ISAAC_HD void clusterBuildFragments(const DevParams &P, const DevReference &R, const u8 *bcl, u32 cluster, const Match *matches, const u64 *offsets,
                                    bool withGaps, bool trim, FragmentWork &work, ClusterFragments &out, Counters &cnt, bool deferSimpleIndels = false)
{
    const u64 begin = offsets[cluster], end = offsets[cluster + 1];
    buildFragments(P, R, bcl + u64(cluster) * P.clusterLength, matches + begin, u32(end - begin), withGaps, trim, work, out, cnt, deferSimpleIndels);
}
ISAAC_HD bool clusterSimpleIndelsPending(const ClusterFragments &f) { return 0 != (f.flags & (CLUSTER_INDEL_PENDING | (CLUSTER_INDEL_PENDING << 1))); }
// the deferred single-indel stage of a cluster (k_indel_fragments)
ISAAC_HD void clusterFinishSimpleIndels(const DevParams &P, const DevReference &R, const u8 *bcl, u32 cluster, FragmentWork &work, ClusterFragments &out, Counters &cnt,
                                        const IndelStage *stage = 0)
{
    finishSimpleIndels(P, R, bcl + u64(cluster) * P.clusterLength, out, work.order, cnt, stage);
}

// second half: the gapped retries' results (in countGappedJobs order) are applied and the lists consolidated.  results == NULL
// with withGaps: the retries run here, one after the other (capacity fallback of the flat pass and the plain serial form).
struct FlatGappedProvider { const GappedResult *next; ISAAC_HD const GappedResult *operator()(u32, u32) { return next++; } };
struct NoGappedProvider { ISAAC_HD const GappedResult *operator()(u32, u32) { return 0; } };
struct SerialGappedProvider
{
    const DevParams *P; const DevReference *R; const u8 *clusterBcl; const ClusterFragments *frags; u32 *tflags; GappedResult result;
    ISAAC_HD const GappedResult *operator()(u32 r, u32 i)
    {
        GappedJob job;
        makeGappedJob(*P, *frags, r, i, 0, job);
        runGappedJobSerial(*P, *R, clusterBcl, job, tflags, result);
        return &result;
    }
};
ISAAC_HD void clusterFinishFragments(const DevParams &P, const DevReference &R, const u8 *bcl, u32 cluster, bool withGaps, const GappedResult *results,
                                     FragmentWork &work, ClusterFragments &out, Counters &cnt)
{
    if (!withGaps) { NoGappedProvider p; finishFragments(P, out, p, work.order, cnt, work.keys, work.keyCap); }
    else if (results) { FlatGappedProvider p; p.next = results; finishFragments(P, out, p, work.order, cnt, work.keys, work.keyCap); }
    else
    {
        SerialGappedProvider p; p.P = &P; p.R = &R; p.clusterBcl = bcl + u64(cluster) * P.clusterLength; p.frags = &out; p.tflags = work.tflags;
        finishFragments(P, out, p, work.order, cnt, work.keys, work.keyCap);
    }
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

// privateCands: room for 2 * PRIVATE_CANDS candidates in the caller's private memory.  Short candidate lists (the usual case) are
// copied there once, 16 bytes at a time; the template logic reads their fields many times over, and private memory is
// interleaved by lane, so a wave's reads of "field f of candidate i" share cache lines instead of touching one line per lane.
#ifndef ISAAC_PRIVATE_CANDS
#define ISAAC_PRIVATE_CANDS 2
#endif
static const u32 PRIVATE_CANDS = ISAAC_PRIVATE_CANDS;
ISAAC_HD void templateCtxInit(TemplateCtx &x, const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, const u8 *bcl, u32 cluster,
                               const ClusterFragments &frags, TemplateWork &work, Counters &cnt, Cand *privateCands = 0)
{
    for (u32 r = 0; r < 2; ++r) { x.cands[r] = frags.cands[r]; x.nCands[r] = frags.nCands[r]; }
    if (privateCands && x.nCands[0] <= PRIVATE_CANDS && x.nCands[1] <= PRIVATE_CANDS)
        for (u32 r = 0; r < 2; ++r)
        {
            for (u32 i = 0; i < x.nCands[r]; ++i) privateCands[r * PRIVATE_CANDS + i] = frags.cands[r][i];
            x.cands[r] = privateCands + r * PRIVATE_CANDS;
        }
    x.P = &P; x.R = &R; x.tls = &tls; x.frags = &frags; x.w = &work; x.cnt = &cnt; x.clusterId = cluster;
    x.rogRead[0] = rog.read[0]; x.rogRead[1] = rog.read[1]; x.rog = rog.pair;
    x.rescueMode = RESCUE_SERIAL; x.jobNext = 0; x.jobCount = 0; x.jobs = 0; x.planWrite = false; x.serialFallbackAllowed = true;
    x.candPositions = 0; x.shadowCands = 0; x.shadowCigars = 0; x.gappedResults = 0; x.gappedJobs = 0; x.candRank = 0; x.sums = 0;
    x.bestRescued = work.shadowList; x.bestRescuedPool = work.shadowCigar;
    x.lanes = 1; x.lane = 0; x.fastSort = false; x.ldsSort = 0; x.ldsSortCap = 0; x.mapqNearInteger = 0;
    for (u32 i = 0; i < 8; ++i) x.prof[i] = 0;
    const u8 *clusterBcl = bcl + u64(cluster) * P.clusterLength;
    for (u32 r = 0; r < 2; ++r)
    {
        x.reads[r].bcl = clusterBcl + P.readOffset[r]; x.reads[r].length = r < P.nReads ? P.readLength[r] : 0;
        x.reads[r].firstCycle = P.firstCycle[r]; x.reads[r].endCyclesMasked = frags.endCyclesMasked[r];
    }
}

// The mate-rescue problems of one cluster, in the order TemplateBuilder would call ShadowAligner::rescueShadow.  The order
// and the windows depend only on the seeded candidates, never on rescue results, so the template logic is simply run with a
// stub rescue.  jobs == NULL: count only.  `cluster` is the index in the tile, `chunkCluster` the index inside the chunk.
ISAAC_HD u32 clusterPlanRescue(const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, double logMismatchQ40,
                               const u8 *bcl, u32 cluster, u32 chunkCluster, const ClusterFragments &frags, TemplateWork &work, RescueJob *jobs, Cand *privateCands = 0)
{
    if (!frags.built) return 0;
    Counters scratch;
    TemplateCtx x;
    templateCtxInit(x, P, R, tls, rog, bcl, cluster, frags, work, scratch, privateCands);
    x.rescueMode = RESCUE_PLAN; x.jobs = jobs; x.planWrite = jobs != 0;
    BamTemplate t;
    work.overflow = 0;
    buildTemplate(x, t, logMismatchQ40);
    if (jobs) for (u32 i = 0; i < x.jobNext; ++i) jobs[i].cluster = chunkCluster;
    return x.jobNext;
}

// ungapped alignment of one rescue candidate position (the body of the loop at ShadowAligner.cpp:206-231)
// summary: the 16 bytes of the result that the plan kernels walk (template.h: CandSummary); its position is left to the caller (rescueSummaryPosition)
ISAAC_HD void rescueAlignCandidate(const DevParams &P, const DevReference &R, const u8 *bcl, u32 cluster, u32 endCyclesMasked, const RescueJob &job, i32 relativePosition,
                                   Cand &out, u32 *cigar3, CandSummary *summary = 0)
{
    ReadView shadowRead;
    const u32 r = job.shadowReadIndex;
    shadowRead.bcl = bcl + u64(cluster) * P.clusterLength + P.readOffset[r]; shadowRead.length = P.readLength[r];
    shadowRead.firstCycle = P.firstCycle[r]; shadowRead.endCyclesMasked = endCyclesMasked;
    CigarPool pool; pool.words = cigar3; pool.used = 0; pool.capacity = 3; pool.overflow = 0;
    // the candidate is made in registers and stored once: initialised in its place and filled in by the scan it reached device memory twice (10.7 M slots a
    // step do not wait in the L2 for the scan to end: 1.56 GB written per step for 0.8 GB of candidates, profiles/r5_final_pmc_summary.json)
#if defined(ISAAC_CAND_IN_PLACE)       // (the form of rounds 1-5, for comparison)
    candInit(out, r);
    out.reverse = job.shadowReverse; out.contigId = job.contigId; out.position = i64(relativePosition) + job.windowBegin;
    alignUngapped(P, R, shadowRead, out, pool, job.adapterRange);
#else
    Cand c;
    candInit(c, r);
    c.reverse = job.shadowReverse; c.contigId = job.contigId; c.position = i64(relativePosition) + job.windowBegin;
    alignUngapped(P, R, shadowRead, c, pool, job.adapterRange);
    out = c;
    if (summary) { summary->logProbability = c.logProbability; summary->mismatchCount = c.mismatchCount; summary->cigarLength = c.cigarLength; summary->relativePosition = 0; }
#endif
}

// an aligned rescue candidate's position relative to its problem's window: the start that was tried + the leading soft clip (the only way an alignment's
// position moves; an unaligned candidate's position is not looked at)
ISAAC_HD i32 rescueSummaryPosition(i32 relativePosition, u32 cigarLength, u32 firstCigarWord)
{ return relativePosition + ((cigarLength && OP_SOFT_CLIP == cigarCode(firstCigarWord)) ? i32(cigarLen(firstCigarWord)) : 0); }

// What the flat kernels hand to clusterSelect: the cluster's jobs and the aligned candidates of the chunk
// sums != NULL: the jobs carry their outcome and the cluster's probability sums are given (RESCUE_PRECOMPUTED, after k_cluster_sums)
struct RescueInputs { RescueJob *jobs; u32 jobCount; const Cand *shadowCands; const u32 *shadowCigars; const u32 *candRank; const GappedResult *gappedResults; const GappedJob *gappedJobs; bool serialFallbackAllowed;
                      const ClusterSums *sums; };
// wave-cooperative execution (k_select_heavy) and the fast probability sort; see TemplateCtx
struct CoopInputs { u32 lanes, lane; bool fastSort; u16 *ldsSort; u32 ldsSortCap; };

// MatchSelector::processMatchList for one cluster: template building, clipping, io::FragmentHeader records.
// records: P.nReads per cluster; cigars: P.nReads * OUT_CIGAR_CAP words per cluster
ISAAC_HD void clusterSelect(const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, double logMismatchQ40,
                            const u8 *bcl, u32 cluster, u32 tile, const ClusterFragments &frags, TemplateWork &work,
                            FragmentRecord *records, u32 *cigars, Counters &cnt, const RescueInputs *rescue = 0, const CoopInputs *coop = 0, Cand *privateCands = 0)
{
    STAMP_BEGIN();
    TemplateCtx x;
    templateCtxInit(x, P, R, tls, rog, bcl, cluster, frags, work, cnt, privateCands);
    if (coop) { x.lanes = coop->lanes; x.lane = coop->lane; x.fastSort = coop->fastSort; x.ldsSort = coop->ldsSort; x.ldsSortCap = coop->ldsSortCap; }
    if (rescue)
    {
        x.rescueMode = RESCUE_LOOKUP; x.jobs = rescue->jobs; x.jobCount = rescue->jobCount; x.shadowCands = rescue->shadowCands; x.shadowCigars = rescue->shadowCigars;
        x.gappedResults = rescue->gappedResults; x.gappedJobs = rescue->gappedJobs; x.candRank = rescue->candRank;
        x.serialFallbackAllowed = rescue->serialFallbackAllowed;
        if (rescue->sums) { x.rescueMode = RESCUE_PRECOMPUTED; x.sums = rescue->sums; }
    }
    BamTemplate t;
    ISAAC_PROF_T0(x);
    STAMP(30);
    const bool store = selectCluster(x, t, logMismatchQ40);
    STAMP(31);
    ISAAC_PROF_ADD(x, 6);
#if defined(ISAAC_PROFILE_HEAVY) && defined(__HIP_DEVICE_COMPILE__)
    if (coop && 0 == x.lane)
        printf("heavy cluster %u jobs %u: copy %lld finish %lld push %lld sortS %lld sortP %lld sums %lld total %lld\n", cluster, x.jobCount,
               x.prof[0], x.prof[1], x.prof[2], x.prof[3], x.prof[4], x.prof[5], x.prof[6]);
#endif
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
        if (x.mapqNearInteger) r.reserved |= RECORD_MAPQ_NEAR_INTEGER;
        r.reserved |= (t.alignmentScore >= 0xffffu ? 0xffffu : t.alignmentScore) << 16;      // BamTemplate::getAlignmentScore for io::getTemplateDuplicateRank (bam_kernels.h)
    }
    STAMP(32);
}

} // namespace isaac
