// The template stage of the main pass, cut to what a cluster's thread still has to do once the flat kernels have run: the mate
// rescues are planned (k_plan_rescue), scanned, aligned, retried and summed elsewhere, so what is left is a walk over the cluster's
// two short candidate lists and a handful of lookups.  The functions here restate TemplateBuilder for exactly that case with as
// little live state as possible: candidates stay where they lie in HBM and are named by list index, a fragment of the template is a
// dozen scalars (LeanFrag), tie lists are a few bytes in a register, and new CIGARs are edited where the record's CIGAR goes anyway.
// Nothing here is indexed at run time in private memory, so the kernels compile without scratch (template.h keeps the general
// form: list-building modes, wave-cooperative execution and the reference's own capacities, for the residual pass and the CPU
// harness).
//
// Behaviour follows (paths relative to /root/reference/src/c++): lib/alignment/TemplateBuilder.cpp:97-174 (buildTemplate),
// :177-226 (getBestFragment), :233-285 (updateMappingScore), :287-391 (locateBestPair), :398-465 (buildPairedEndTemplate),
// :467-493 / :1010-1033 (flagDodgyTemplate), :495-676 (rescueShadow), :716-866 (buildDisjoinedTemplate), :868-1008
// (scoreDisjoinedTemplate), :1035-1058 (pickBestFragment); lib/alignment/BamTemplate.cpp:47-72; lib/alignment/ShadowAligner.cpp:119-197;
// lib/alignment/matchSelector/SemialignedEndsClipper.cpp:31-205, OverlappingEndsClipper.cpp:46-183; include/io/Fragment.hh:101-246;
// lib/alignment/MatchSelector.cpp:296-366.
#pragma once
#include "template.h"

namespace isaac
{

// one end of the template being built: the FragmentMetadata fields that the scores, the clippers and the record read or change
struct LeanFrag
{
    i64 position; const u32 *cigar;      // cigar: the fragment's own first word (pool + cigarOffset)
    u32 contigId, observedLength, alignmentScore;
    u16 cigarLength, editDistance, lowClipped, highClipped, gapCount;
    u8 reverse, readIndex;
    // while the end clippers work (leanLoadEdges .. leanWriteCigar): the first two and the last two words of the CIGAR, and the bases its first and its
    // last ALIGN operation have lost to a soft clip since.  The CIGAR itself stays where it is and is written once, clipped, into the record's slot
    // (round 4 edited it in the slot: every look at a word behind an edit was a round trip to memory)
    u32 w0, w1, wl, wp; u16 frontClip, backClip;
};
ISAAC_HD void leanInit(LeanFrag &f, u32 readIndex)
{
    f.position = 0; f.cigar = 0; f.contigId = MAX_CONTIG_ID; f.observedLength = 0; f.alignmentScore = 0xffffffffu; f.cigarLength = 0; f.editDistance = 0;
    f.lowClipped = 0; f.highClipped = 0; f.gapCount = 0; f.reverse = 0; f.readIndex = u8(readIndex);
    f.w0 = f.w1 = f.wl = f.wp = 0; f.frontClip = f.backClip = 0;
}
ISAAC_HD void leanLoad(LeanFrag &f, const Cand &c, const u32 *pool)
{
    f.position = c.position; f.cigar = pool + c.cigarOffset; f.contigId = c.contigId; f.observedLength = c.observedLength; f.alignmentScore = c.alignmentScore;
    f.cigarLength = c.cigarLength; f.editDistance = c.editDistance; f.lowClipped = c.lowClipped; f.highClipped = c.highClipped; f.gapCount = c.gapCount;
    f.reverse = c.reverse; f.readIndex = c.readIndex;
    f.w0 = f.w1 = f.wl = f.wp = 0; f.frontClip = f.backClip = 0;
}
// c ? a : b, field by field: a conditional on the structs themselves is a choice between two addresses, and a fragment whose address
// is chosen at run time lives in scratch memory instead of registers
ISAAC_HD LeanFrag leanPick(bool c, const LeanFrag &a, const LeanFrag &b)
{
    LeanFrag f;
    f.position = c ? a.position : b.position; f.cigar = c ? a.cigar : b.cigar; f.contigId = c ? a.contigId : b.contigId; f.observedLength = c ? a.observedLength : b.observedLength;
    f.alignmentScore = c ? a.alignmentScore : b.alignmentScore; f.cigarLength = c ? a.cigarLength : b.cigarLength; f.editDistance = c ? a.editDistance : b.editDistance;
    f.lowClipped = c ? a.lowClipped : b.lowClipped; f.highClipped = c ? a.highClipped : b.highClipped; f.gapCount = c ? a.gapCount : b.gapCount;
    f.reverse = c ? a.reverse : b.reverse; f.readIndex = c ? a.readIndex : b.readIndex;
    f.w0 = c ? a.w0 : b.w0; f.w1 = c ? a.w1 : b.w1; f.wl = c ? a.wl : b.wl; f.wp = c ? a.wp : b.wp; f.frontClip = c ? a.frontClip : b.frontClip; f.backClip = c ? a.backClip : b.backClip;
    return f;
}
ISAAC_HD ReadView leanPickRead(bool c, const ReadView &a, const ReadView &b)
{ ReadView r; r.bcl = c ? a.bcl : b.bcl; r.length = c ? a.length : b.length; r.endCyclesMasked = c ? a.endCyclesMasked : b.endCyclesMasked; r.firstCycle = c ? a.firstCycle : b.firstCycle; return r; }
ISAAC_HD bool leanAligned(const LeanFrag &f) { return 0 != f.cigarLength; }
ISAAC_HD u32 leanObservedLength(const LeanFrag &f) { return leanAligned(f) ? f.observedLength : 0; }
ISAAC_HD void leanSetNoMatch(LeanFrag &f) { f.cigarLength = 0; f.alignmentScore = 0xffffffffu; f.contigId = MAX_CONTIG_ID; f.position = 0; }
ISAAC_HD u64 leanFStrandPos(const LeanFrag &f) { return MAX_CONTIG_ID == f.contigId ? REFPOS_NOMATCH : refpos(f.contigId, u64(f.position)); }

// TemplateLengthStatistics::checkModel on two template ends (template.h: tlsCheckModel)
ISAAC_HD u32 leanCheckModel(const DevTls &t, const LeanFrag &f1, const LeanFrag &f2)
{
    // (every field is read into a scalar before anything is chosen: "the observed length of whichever starts first" through a chosen
    // address would put both fragments into scratch memory)
    const i64 p1 = f1.position, p2 = f2.position; const i64 o1 = i64(leanObservedLength(f1)), o2 = i64(leanObservedLength(f2));
    if (f1.contigId != f2.contigId) return TLS_NOMATCH;
    const i32 model = i32(((p1 <= p2) ? 0 : 4) | (f1.reverse ? 2 : 0) | (f2.reverse ? 1 : 0));
    if (model != t.bestModel[0] && model != t.bestModel[1]) return TLS_NOMATCH;
    const u64 length = (p1 < p2) ? u64(imax<i64>(p2 + o2 - p1, o1)) : u64(imax<i64>(p1 + o1 - p2, o2));
    return (length > t.max) ? TLS_OVERSIZED : (length < t.min) ? TLS_UNDERSIZED : TLS_NOMINAL;
}

struct LeanTemplate { LeanFrag f0, f1; u32 n, alignmentScore; bool properPair; };

// what the lean functions know about the cluster
struct LeanCtx
{
    const DevParams *P; const DevReference *R; const DevTls *tls;
    const Cand *l0, *l1; u32 n0, n1;          // the candidate lists
    const u32 *pool;                          // the cluster's CIGAR words (Cand::cigarOffset is relative to it)
    double rogRead0, rogRead1, rog, logMismatchQ40;
    u32 clusterId;
    u32 mapqNearInteger;
};
ISAAC_HD const Cand *leanList(const LeanCtx &x, u32 r) { return r ? x.l1 : x.l0; }
ISAAC_HD u32 leanCount(const LeanCtx &x, u32 r) { return r ? x.n1 : x.n0; }
ISAAC_HD double leanRogRead(const LeanCtx &x, u32 r) { return r ? x.rogRead1 : x.rogRead0; }

ISAAC_HD u32 leanMapq(LeanCtx &x, double ratio)
{   // template.h: mapqFloor
    const double v = -10.0 * log10(ratio);
    const double fl = floor(v);
    const double d = v - fl;
#if defined(ISAAC_TEST_MAPQ_SKEW) && defined(__HIP_DEVICE_COMPILE__)
    // a test build (tests/test_gpu_parity.py::test_flagged_clusters_are_resolved_on_the_host): a twentieth of all values counts as "near an integer", and the device gets
    // every one of them wrong by one -- what isaac_gpu_resolve_flagged must put right
    if (d > 0.975 || d < 0.025) { ++x.mapqNearInteger; return u32(fl) + 1; }
#endif
    if (d > 1.0 - 1e-11 || (d < 1e-11 && fl >= 1.0)) ++x.mapqNearInteger;
    return u32(fl);
}

// getBestFragment (TemplateBuilder.cpp:177-226)
ISAAC_HD u32 leanBestFragment(const LeanCtx &x, u32 r)
{
    const Cand *list = leanList(x, r); const u32 n = leanCount(x, r);
    u32 bestScore = 0xffffffffu; double bestLp = -1.7976931348623157e308;
    u32 first = 0, count = 0;
    for (u32 i = 0; i < n; ++i)
    {
        const u32 sws = list[i].smithWatermanScore; const double lp = list[i].logProbability;
        if (bestScore > sws || (bestScore == sws && lpLess(bestLp, lp))) { bestScore = sws; bestLp = lp; first = i; count = 1; }
        else if (bestScore == sws && lpEquals(bestLp, lp)) ++count;
    }
    if (!x.P->scatterRepeats || count < 2) return first;
    u32 want = x.clusterId % count;
    if (!want) return first;
    for (u32 i = first + 1; i < n; ++i)
        if (bestScore == list[i].smithWatermanScore && lpEquals(bestLp, list[i].logProbability) && 0 == --want) return i;
    return first;
}

// the probability part of updateMappingScore (TemplateBuilder.cpp:233-285): the MAPQ of candidate listIndex of read r among its list
ISAAC_HD u32 leanListMapq(LeanCtx &x, u32 r, u32 listIndex)
{
    const Cand *list = leanList(x, r); const u32 n = leanCount(x, r);
    double neighborProbability = leanRogRead(x, list[listIndex].readIndex);
    for (u32 i = 0; i < n; ++i) if (listIndex != i) neighborProbability += exp(list[i].logProbability);
    return leanMapq(x, neighborProbability / (neighborProbability + exp(list[listIndex].logProbability)));
}
ISAAC_HD bool leanUpdateMappingScore(LeanCtx &x, LeanFrag &fragment, bool wellAnchored, u32 r, u32 listIndex, bool forceWellAnchored)
{
    if (forceWellAnchored || wellAnchored) { fragment.alignmentScore = leanListMapq(x, r, listIndex); return true; }
    fragment.alignmentScore = 0;
    return false;
}

// BestPairInfo of locateBestPair without its lists: the ties' list indexes a byte each, the first LEAN_TIES of them
static const u32 LEAN_TIES = 4;
static const u32 LEAN_NO_TIE = 0xffffffffu;
struct LeanBestPair
{
    double bestLp, total; u64 bestScore; u32 resolved, editDistance;
    u32 ties0, ties1, nTies;          // list indexes of the equally good pairs, byte k = tie k (the first LEAN_TIES), nTies: all of them
    u32 want0, want1;                 // the tie asked for by index (leanLocateBestPair's `want`)
};
ISAAC_HD u32 leanTie(u32 packed, u32 k) { return (packed >> (8 * k)) & 0xffu; }

// locateBestPair (TemplateBuilder.cpp:287-391).  withProbabilities: the sum of the pair probabilities is wanted (k_select), not only
// which pair is best (k_plan_rescue).
// want: index of a tie to report in want0 / want1 whatever its rank (a second pass for --scatter-repeats when the tie it picks is not among the kept ones)
template <bool withProbabilities>
ISAAC_HD void leanLocateBestPair(const LeanCtx &x, LeanBestPair &ret, u32 want = LEAN_NO_TIE)
{
    const Cand *l0 = x.l0, *l1 = x.l1;
    const u32 n0 = x.n0, n1 = x.n1;
    const DevTls &tls = *x.tls;
    ret.bestLp = -1.7976931348623157e308; ret.bestScore = ~u64(0); ret.resolved = 0; ret.editDistance = 0; ret.total = 0.0;
    ret.ties0 = 0; ret.ties1 = 0; ret.nTies = 1;     // BestPairInfo::init(0, 0)
    ret.want0 = 0; ret.want1 = 0;
    u32 b0 = 0, b1 = 0;
    while (b0 != n0 && b1 != n1)
    {
        const u32 c0 = l0[b0].contigId, c1 = l1[b1].contigId;
        u32 e0 = b0 + 1; while (e0 != n0 && l0[e0].contigId == c0) ++e0;
        u32 e1 = b1 + 1; while (e1 != n1 && l1[e1].contigId == c1) ++e1;
        if (c0 == c1)
        {
            for (u32 i = b0; i != e0; ++i)
            {
                const Cand &a = l0[i];
                const i64 aPos = a.position; const u32 aObs = candObservedLength(a); const bool aRev = a.reverse; const double aLp = a.logProbability; const u32 aSws = a.smithWatermanScore;
                for (u32 j = b1; j != e1; ++j)
                {
                    const Cand &b = l1[j];
                    const i64 bPos = b.position; const u32 bObs = candObservedLength(b);
                    // tlsMatchModel
                    const u64 length = (aPos < bPos) ? u64(imax<i64>(bPos + i64(bObs) - aPos, i64(aObs))) : u64(imax<i64>(aPos + i64(aObs) - bPos, i64(bObs)));
                    const i32 model = i32(((aPos <= bPos) ? 0 : 4) | (aRev ? 2 : 0) | (b.reverse ? 1 : 0));
                    if (!((length <= u64(tls.max + TEMPLATE_LENGTH_THRESHOLD)) && (model == tls.bestModel[0] || model == tls.bestModel[1]))) continue;
                    const double lp = aLp + b.logProbability;
                    const u64 templateScore = u64(aSws + b.smithWatermanScore);
                    if (withProbabilities) ret.total += exp(lp);
                    if (0 == ret.resolved || ret.bestScore > templateScore || (templateScore == ret.bestScore && lpLess(ret.bestLp, lp)))
                    { ret.ties0 = i; ret.ties1 = j; ret.nTies = 1; ret.bestScore = templateScore; ret.bestLp = lp; if (0 == want) { ret.want0 = i; ret.want1 = j; } }
                    else if (templateScore == ret.bestScore && lpEquals(lp, ret.bestLp))
                    {
                        if (ret.nTies < LEAN_TIES) { ret.ties0 |= i << (8 * ret.nTies); ret.ties1 |= j << (8 * ret.nTies); }
                        if (ret.nTies == want) { ret.want0 = i; ret.want1 = j; }
                        ++ret.nTies;
                    }
                    ++ret.resolved;
                }
            }
            b0 = e0; b1 = e1;
        }
        else if (c0 < c1) b0 = e0; else b1 = e1;
    }
    if (ret.resolved) ret.editDistance = u32(l0[leanTie(ret.ties0, 0)].editDistance) + u32(l1[leanTie(ret.ties1, 0)].editDistance);
}

// ShadowAligner::rescueShadow's first half for one orphan (template.h: planRescue), from the fields it reads
struct LeanOrphan { i64 position; u32 contigId, observedLength; u8 reverse, readIndex, aligned, listIndex; };
ISAAC_HD void leanPlanRescue(const LeanCtx &x, const LeanOrphan &orphan, i64 bestTemplateLength, u32 chunkCluster, RescueJob &job)
{
    const DevTls &tls = *x.tls;
    job.windowBegin = 0; job.windowLen = 0; job.cluster = chunkCluster; job.contigId = orphan.contigId; job.candBase = 0; job.nCands = 0; job.pushes = 0;
    job.bitmapBase = 0; job.bitmapWords = 0; job.valid = 0; job.fallback = 0; job.gappedBase = 0xffffffffu; job.nGapped = 0; job.nAligned = 0; job.bestRank = 0; job.bestSlot = 0; job.lastAligned = 0;
    job.adapterRange = 0; job.pad = 0; job.take = 0; job.finalBestRank = 0; job.finalBestSlot = 0; job.finalBestGapped = 0xffffffffu; job.rescued = 0; job.windowBaseHigh = 0; job.windowBaseLow = 0;
    job.orphanListIndex = orphan.listIndex;
    job.shadowReadIndex = u8((orphan.readIndex + 1) % 2);
    job.shadowReverse = 0;
    if (!tlsIsCoherent(tls)) return;
    job.shadowReverse = u8(tlsMateOrientation(tls, orphan.readIndex, orphan.reverse));
    // calculateShadowRescueRange (ShadowAligner.cpp:119-149)
    const u32 readLength0 = x.P->readLength[0], readLength1 = x.P->nReads > 1 ? x.P->readLength[1] : 0;
    const u32 orphanLength = orphan.readIndex ? readLength1 : readLength0, shadowLength = orphan.readIndex ? readLength0 : readLength1;
    // TemplateLengthStatistics::mateMinPosition / mateMaxPosition (template.h: tlsMateMinPosition, tlsMateMaxPosition)
    i64 shadowMinPosition = orphan.position, shadowMaxPosition = orphan.position;
    if (tlsIsValidModel(tls, orphan.reverse, orphan.readIndex))
    {
        if (tlsFirstFragment(tls, orphan.reverse, orphan.readIndex)) { shadowMinPosition = orphan.position + i64(tls.mateMin) - i64(shadowLength); shadowMaxPosition = orphan.position + i64(tls.mateMax) - i64(shadowLength); }
        else { shadowMinPosition = orphan.position - i64(tls.mateMax) + i64(orphanLength); shadowMaxPosition = orphan.position - i64(tls.mateMin) + i64(orphanLength); }
    }
    shadowMaxPosition += i64(shadowLength) - 1;
    if (bestTemplateLength)
    {
        const u32 observed = orphan.aligned ? orphan.observedLength : 0;
        const i64 fpos = MAX_CONTIG_ID == orphan.contigId ? i64(refposPosition(REFPOS_NOMATCH)) : i64(refposPosition(refpos(orphan.contigId, u64(orphan.position))));
        const i64 rpos = MAX_CONTIG_ID == orphan.contigId ? i64(refposPosition(REFPOS_NOMATCH))
                                                           : i64(refposPosition(refpos(orphan.contigId, u64(imax<i64>(orphan.position + i64(observed), 1) - 1))));
        if (shadowMinPosition < fpos) shadowMinPosition = imin(rpos - bestTemplateLength, shadowMinPosition);
        if (shadowMaxPosition > fpos) shadowMaxPosition = imax(fpos + bestTemplateLength, shadowMaxPosition);
    }
    const i64 rangeFirst = shadowMinPosition - 10, rangeSecond = shadowMaxPosition + 10;
    if (rangeSecond < rangeFirst) return;
    if (rangeSecond + 1 + i64(shadowLength) < 0) return;
    const i64 referenceSize = i64(contigLength(*x.R, orphan.contigId));
    job.windowBegin = imax<i64>(0, rangeFirst);
    const i64 windowEnd = imin(referenceSize, rangeSecond + 1);
    job.windowLen = windowEnd > job.windowBegin ? u32(windowEnd - job.windowBegin) : 0;
    const u64 windowBase = x.R->contigOffset[orphan.contigId] + u64(job.windowBegin);
    job.windowBaseHigh = u16(windowBase >> 32); job.windowBaseLow = u32(windowBase);
    job.valid = 1;
}
ISAAC_HD LeanOrphan leanOrphan(const Cand &c, u32 listIndex)
{
    LeanOrphan o; o.position = c.position; o.contigId = c.contigId; o.observedLength = c.observedLength; o.reverse = c.reverse; o.readIndex = c.readIndex;
    o.aligned = u8(candAligned(c)); o.listIndex = u8(listIndex);
    return o;
}

// BestPairInfo::getBestTemplateLength (TemplateBuilder.hh:276-287) for the pair (l0[i0], l1[i1])
ISAAC_HD i64 leanTemplateLength(const LeanCtx &x, u32 i0, u32 i1)
{
    const Cand &a = x.l0[i0], &c = x.l1[i1];
    const u64 templateStart = imin(candFStrandPos(a), candFStrandPos(c));
    const u64 templateEnd = imax(candRStrandPos(a), candRStrandPos(c));
    return i64(refposPosition(templateEnd)) - i64(refposPosition(templateStart));
}

// which pair of a LeanBestPair buildPairedEndTemplate takes (TemplateBuilder.cpp:404-413): the first, or with --scatter-repeats the
// (cluster % ties)-th -- found by a second walk over the pairs when it is not among the ones kept
ISAAC_HD void leanChosenPair(const LeanCtx &x, const LeanBestPair &b, u32 &i0, u32 &i1)
{
    const u32 repeatIndex = x.P->scatterRepeats ? x.clusterId % b.nTies : 0;
    if (repeatIndex < LEAN_TIES) { i0 = leanTie(b.ties0, repeatIndex); i1 = leanTie(b.ties1, repeatIndex); return; }
    LeanBestPair again;
    leanLocateBestPair<false>(x, again, repeatIndex);
    i0 = again.want0; i1 = again.want1;
}

// The mate-rescue problems of one cluster in the order TemplateBuilder would pose them (template.h / cluster_ops.h: clusterPlanRescue,
// i.e. buildTemplate with a stub rescue): which orphans are rescued and where depends on the seeded candidates alone.
// jobs == NULL: count only.  Returns the number of problems.
ISAAC_HD u32 leanPlanCluster(LeanCtx &x, u32 chunkCluster, RescueJob *jobs)
{
    u32 nJobs = 0;
    const u32 n0 = x.n0, n1 = x.n1;
    if (2 != x.P->nReads) return 0;
    if (n0 && n1)
    {
        LeanBestPair bc;
        leanLocateBestPair<false>(x, bc);
        u32 i0 = 0, i1 = 0;
        if (bc.resolved)
        {
            leanChosenPair(x, bc, i0, i1);
            const Cand &r1 = x.l0[i0], &r2 = x.l1[i1];
            // buildPairedEndTemplate's return value: both reads count as well anchored as soon as one is
            const bool ok = (candWellAnchored(r1) || candWellAnchored(r2)) && !r1.repeatSeedsCount && !r2.repeatSeedsCount;
            if (ok && !bc.editDistance) return 0;
        }
        // buildDisjoinedTemplate (TemplateBuilder.cpp:716-866)
        const u32 bd0 = leanBestFragment(x, 0), bd1 = leanBestFragment(x, 1);
        const i64 bestTemplateLength = bc.resolved ? leanTemplateLength(x, i0, i1) : 0;
        for (u32 orphanIndex = 0; 2 > orphanIndex; ++orphanIndex)
        {
            const Cand *orphans = leanList(x, orphanIndex); const u32 nOrphans = leanCount(x, orphanIndex);
            const double bestLp = orphans[orphanIndex ? bd1 : bd0].logProbability;
            for (u32 oi = 0; oi < nOrphans; ++oi)
            {
                const Cand &orphan = orphans[oi];
                const bool skip = bc.resolved ? u32(orphan.editDistance) > (bc.editDistance + SKIP_ORPHAN_EDIT_DISTANCE) : lpLess(orphan.logProbability + 100.0, bestLp);
                if (skip) continue;
                if (jobs) leanPlanRescue(x, leanOrphan(orphan, oi), bestTemplateLength, chunkCluster, jobs[nJobs]);
                ++nJobs;
            }
        }
    }
    else if (n0 || n1)
    {   // TemplateBuilder::rescueShadow (TemplateBuilder.cpp:495-676)
        const u32 orphanIndex = n0 ? 0 : 1;
        const Cand *orphans = leanList(x, orphanIndex); const u32 nOrphans = leanCount(x, orphanIndex);
        const double bestLp = orphans[leanBestFragment(x, orphanIndex)].logProbability;
        for (u32 oi = 0; oi < nOrphans; ++oi)
        {
            const Cand &orphan = orphans[oi];
            if (lpLess(orphan.logProbability + 100.0, bestLp)) continue;
            if (jobs) leanPlanRescue(x, leanOrphan(orphan, oi), 0, chunkCluster, jobs[nJobs]);
            ++nJobs;
        }
    }
    return nJobs;
}

// ------------------------------------------------------------------------------------------------------------------
// k_select

// what the flat kernels left for the cluster: its rescue problems with their outcome, the best shadows, the sums
struct LeanRescue
{
    const RescueJob *jobs; u32 jobCount; const Cand *shadowCands; const u32 *shadowCigars; const GappedResult *gappedResults; const ClusterSums *sums;
    // RescueJob::out of problem k at outcomes + k * outcomeStride bytes: in the records themselves (stride sizeof(RescueJob)), or a copy the
    // kernel made next to its copy of the candidates (k_select: stride sizeof(RescueOutcome))
    const u8 *outcomes; u32 outcomeStride;
};
ISAAC_HD const RescueOutcome &leanOutcome(const LeanRescue &rs, u32 k) { return *reinterpret_cast<const RescueOutcome *>(rs.outcomes + size_t(k) * rs.outcomeStride); }
ISAAC_HD void leanOutcomesInPlace(LeanRescue &rs) { rs.outcomes = reinterpret_cast<const u8 *>(rs.jobs) + offsetof(RescueJob, out); rs.outcomeStride = u32(sizeof(RescueJob)); }
// the best shadow of a successful rescue (template.h: shadowRescue, RESCUE_PRECOMPUTED): where it lies and where its CIGAR words are
struct LeanShadow { const Cand *c; const u32 *cigar; u32 cigarLength; };
ISAAC_HD LeanShadow leanShadowOf(const LeanRescue &rs, const RescueJob &job)
{
    LeanShadow s;
    if (0xffffffffu != job.finalBestGapped) { const GappedResult &g = rs.gappedResults[job.finalBestGapped]; s.c = &g.out; s.cigar = g.cigar; s.cigarLength = u16(g.nCigar); }
    else { s.c = rs.shadowCands + job.finalBestSlot; s.cigar = rs.shadowCigars + u64(job.finalBestSlot) * 3; s.cigarLength = s.c->cigarLength; }
    return s;
}
ISAAC_HD void leanLoadShadow(LeanFrag &f, const LeanShadow &s)
{
    Cand const &c = *s.c;
    f.position = c.position; f.cigar = s.cigar; f.contigId = c.contigId; f.observedLength = c.observedLength; f.alignmentScore = c.alignmentScore;
    f.cigarLength = u16(s.cigarLength); f.editDistance = c.editDistance; f.lowClipped = c.lowClipped; f.highClipped = c.highClipped; f.gapCount = c.gapCount;
    f.reverse = c.reverse; f.readIndex = c.readIndex;
    f.w0 = f.w1 = f.wl = f.wp = 0; f.frontClip = f.backClip = 0;
}
// isVeryBadAlignment (TemplateBuilder.cpp:52-62) of a rescued shadow / of a list candidate
ISAAC_HD bool leanVeryBad(const Cand &f, const u32 *cigar, u32 cigarLength, double logMismatchQ40)
{
    u32 mapped = 0;
    for (u32 i = 0; i < cigarLength; ++i) if (OP_ALIGN == cigarCode(cigar[i])) mapped += cigarLen(cigar[i]);
    return f.matchesInARow < 32 && (u32(f.mismatchCount) > mapped / 8 || f.logProbability < logMismatchQ40 / 4 * mapped);
}
ISAAC_HD bool leanVeryBad(const RescueOutcome &o, double logMismatchQ40)
{
    const u32 mapped = o.mapped;
    return o.matchesInARow < 32 && (u32(o.mismatchCount) > mapped / 8 || o.logProbability < logMismatchQ40 / 4 * mapped);
}

// the equally good rescued pairs of one orphan side: list index of the orphan and index of its rescue problem, a byte each
struct LeanRescuedBest
{
    double bestLp, total; u64 bestScore; u32 resolved;
    u32 orphans, jobs, n;        // byte k: tie k (the first LEAN_TIES); n: ties of the side the best pair came from
    u32 overflow;
};
ISAAC_HD void leanRescuedClear(LeanRescuedBest &b) { b.bestLp = -1.7976931348623157e308; b.bestScore = ~u64(0); b.resolved = 0; b.total = 0.0; b.orphans = 0; b.jobs = 0; b.n = 0; b.overflow = 0; }

ISAAC_HD bool leanFlagDodgy(const LeanCtx &x, LeanFrag &orphan, LeanFrag &shadow, LeanTemplate &t)
{
    if (-1 == x.P->dodgyAlignmentScore) { leanSetNoMatch(orphan); leanSetNoMatch(shadow); t.alignmentScore = 0xffffffffu; return false; }
    orphan.alignmentScore = 0xffffffffu; shadow.alignmentScore = 0xffffffffu; t.alignmentScore = 0xffffffffu;
    return true;
}
ISAAC_HD void leanClampDodgy(u32 &a) { a = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, a); }
ISAAC_HD void leanPut(LeanTemplate &t, u32 r, const LeanFrag &f) { t.f0 = leanPick(0 == r, f, t.f0); t.f1 = leanPick(0 != r, f, t.f1); }

// one orphan's rescue outcome against the running best (the bodies of the loops at TemplateBuilder.cpp:527-590 and :742-824)
// knownResolved / knownEditDistance: buildDisjoinedTemplate's extra condition; pass 0 / 0 from rescueShadow
ISAAC_HD void leanConsiderRescued(const LeanCtx &x, const LeanRescue &rs, const Cand &orphan, u32 oi, u32 jobIndex, bool sideIsBest, bool disjoined, u32 knownResolved, u32 knownEditDistance,
                                  LeanRescuedBest &best, bool &newBest)
{
    newBest = false;
    // (the problem's best shadow through the summary finishRescueFlat left in the record: following the record to the shadow and its CIGAR
    // is two more memory round trips per orphan, one orphan after the other)
    const RescueOutcome &bestRescued = leanOutcome(rs, jobIndex);
    if (!bestRescued.rescued) return;
    const double lp = orphan.logProbability + bestRescued.logProbability;
    if (leanVeryBad(bestRescued, x.logMismatchQ40)) return;
    if (disjoined && knownResolved && !((knownEditDistance + SKIP_ORPHAN_EDIT_DISTANCE) >= u32(orphan.editDistance) + u32(bestRescued.editDistance))) return;
    const u64 templateScore = u64(orphan.smithWatermanScore + bestRescued.smithWatermanScore);
    if (0 == best.resolved || templateScore < best.bestScore || (templateScore == best.bestScore && lpLess(best.bestLp, lp)))
    {
        best.bestLp = lp; best.bestScore = templateScore; best.orphans = oi; best.jobs = jobIndex; best.n = 1; newBest = true;
    }
    else if (templateScore == best.bestScore && lpEquals(lp, best.bestLp))
    {
        // ties of the other side than the best pair's go to lists nothing reads
        if (sideIsBest)
        {
            if (best.n < LEAN_TIES) { best.orphans |= oi << (8 * best.n); best.jobs |= jobIndex << (8 * best.n); }
            ++best.n;
        }
    }
    ++best.resolved;
}

// TemplateBuilder::rescueShadow (TemplateBuilder.cpp:495-676): exactly one read has candidates
ISAAC_HD bool leanRescueShadow(LeanCtx &x, const LeanRescue &rs, LeanTemplate &t, u32 &overflow)
{
    const u32 orphanIndex = x.n0 ? 0 : 1;
    const u32 shadowIndex = 1 - orphanIndex;
    const Cand *orphans = leanList(x, orphanIndex); const u32 nOrphans = leanCount(x, orphanIndex);
    const u32 bestOrphanIt = leanBestFragment(x, orphanIndex);
    const double bestOrphanLp = orphans[bestOrphanIt].logProbability;
    LeanRescuedBest bp; leanRescuedClear(bp);
    u32 jobNext = 0;
    for (u32 oi = 0; oi < nOrphans; ++oi)
    {
        const Cand &orphan = orphans[oi];
        if (lpLess(orphan.logProbability + 100.0, bestOrphanLp)) continue;
        bool newBest;
        leanConsiderRescued(x, rs, orphan, oi, jobNext++, true, false, 0, 0, bp, newBest);
    }
    bp.total = rs.sums->ordered;
    bool ret = true;
    LeanFrag orphanF, shadowF;
    if (0 < bp.resolved)
    {
        const double totalShadowProbability = rs.sums->shadow[orphanIndex];
        const u32 repeatIndex = x.P->scatterRepeats ? x.clusterId % bp.n : 0;
        if (repeatIndex >= LEAN_TIES) { overflow = 1; return false; }
        const u32 listIdx = leanTie(bp.orphans, repeatIndex);
        const Cand &orphanC = orphans[listIdx];
        leanLoad(orphanF, orphanC, x.pool);
        const LeanShadow s = leanShadowOf(rs, rs.jobs[leanTie(bp.jobs, repeatIndex)]);
        leanLoadShadow(shadowF, s);
        const double shadowLp = s.c->logProbability;
        const bool orphanWellAnchored = candWellAnchored(orphanC);
        const bool assumeWellAnchored = leanUpdateMappingScore(x, orphanF, orphanWellAnchored, orphanIndex, listIdx, 0 == u32(orphanF.editDistance) + u32(shadowF.editDistance));
        if (assumeWellAnchored)
        {
            const double shadowRog = leanRogRead(x, shadowF.readIndex);
            const double otherShadowsProbability = (totalShadowProbability - exp(shadowLp)) + shadowRog;
            shadowF.alignmentScore = leanMapq(x, otherShadowsProbability / (totalShadowProbability + shadowRog));
            const double otherPairsProbability = (bp.total - exp(bp.bestLp)) + x.rog;
            t.alignmentScore = leanMapq(x, otherPairsProbability / (bp.total + x.rog));
            if (!orphanF.alignmentScore || !orphanWellAnchored) { leanClampDodgy(t.alignmentScore); leanClampDodgy(shadowF.alignmentScore); leanClampDodgy(orphanF.alignmentScore); }
        }
        else ret = leanFlagDodgy(x, orphanF, shadowF, t);
        t.properPair = TLS_NOMINAL == leanCheckModel(*x.tls, orphanF, shadowF);
    }
    else
    {
        const Cand &orphanC = orphans[bestOrphanIt];
        leanLoad(orphanF, orphanC, x.pool);
        leanInit(shadowF, shadowIndex);
        if (leanVeryBad(orphanC, orphanF.cigar, orphanF.cigarLength, x.logMismatchQ40)) { leanSetNoMatch(orphanF); leanSetNoMatch(shadowF); ret = false; }
        else
        {
            shadowF.contigId = orphanF.contigId; shadowF.position = orphanF.position; shadowF.alignmentScore = 0; shadowF.cigarLength = 0;
            const bool orphanWellAnchored = candWellAnchored(orphanC);
            if (!leanUpdateMappingScore(x, orphanF, orphanWellAnchored, orphanIndex, bestOrphanIt, 0 == orphanF.editDistance)) ret = leanFlagDodgy(x, orphanF, shadowF, t);
            else
            {
                if (!orphanWellAnchored) leanClampDodgy(orphanF.alignmentScore);
                t.alignmentScore = 0;
            }
        }
    }
    leanPut(t, orphanIndex, orphanF); leanPut(t, shadowIndex, shadowF);
    return ret;
}

// buildDisjoinedTemplate + scoreDisjoinedTemplate (TemplateBuilder.cpp:716-1008).  known: locateBestPair's result, (k0, k1) the pair
// buildPairedEndTemplate took of it (its fragments are in t when known.resolved)
ISAAC_HD bool leanDisjoinedTemplate(LeanCtx &x, const LeanRescue &rs, LeanTemplate &t, const LeanBestPair &known, u32 k0, u32 k1, u32 &overflow)
{
    const u32 bd0 = leanBestFragment(x, 0), bd1 = leanBestFragment(x, 1);
    LeanRescuedBest bo; leanRescuedClear(bo);
    u32 bestOrphanIndex = 0;
    u32 jobNext = 0;
    for (u32 orphanIndex = 0; 2 > orphanIndex; ++orphanIndex)
    {
        const Cand *orphans = leanList(x, orphanIndex); const u32 nOrphans = leanCount(x, orphanIndex);
        const double bestLp = orphans[orphanIndex ? bd1 : bd0].logProbability;
        for (u32 oi = 0; oi < nOrphans; ++oi)
        {
            const Cand &orphan = orphans[oi];
            const bool skip = known.resolved ? u32(orphan.editDistance) > (known.editDistance + SKIP_ORPHAN_EDIT_DISTANCE) : lpLess(orphan.logProbability + 100.0, bestLp);
            if (skip) continue;
            bool newBest;
            leanConsiderRescued(x, rs, orphan, oi, jobNext++, bestOrphanIndex == orphanIndex && bo.resolved, true, known.resolved, known.editDistance, bo, newBest);
            if (newBest) bestOrphanIndex = orphanIndex;
        }
    }
    bool ret = true;
    if (0 < bo.resolved)
    {
        const u32 bestShadowIndex = 1 - bestOrphanIndex;
        const double totalShadowProbability = rs.sums->shadow[bestOrphanIndex];
        const double totalOrphanProbability = rs.sums->shadow[bestShadowIndex];
        bo.total += rs.sums->pair;
        const u32 repeatIndex = x.P->scatterRepeats ? x.clusterId % bo.n : 0;
        if (repeatIndex >= LEAN_TIES) { overflow = 1; return false; }
        const u32 orphanListIdx = leanTie(bo.orphans, repeatIndex);
        const Cand *orphanList = leanList(x, bestOrphanIndex), *shadowList = leanList(x, bestShadowIndex);
        const Cand &bestOrphan = orphanList[orphanListIdx];
        const LeanShadow s = leanShadowOf(rs, rs.jobs[leanTie(bo.jobs, repeatIndex)]);
        const Cand &bestShadowC = *s.c;
        const u32 knownOrphan = bestOrphanIndex ? k1 : k0, knownShadow = bestOrphanIndex ? k0 : k1;
        const bool rediscovered = !repeatIndex && known.resolved && candEqual(orphanList[knownOrphan], bestOrphan) && candEqual(shadowList[knownShadow], bestShadowC);
        LeanFrag orphanF, shadowF;
        leanLoad(orphanF, bestOrphan, x.pool);
        leanLoadShadow(shadowF, s);
        const bool shadowWellAnchored = rediscovered && candWellAnchored(shadowList[knownShadow]);
        const bool orphanWellAnchored = candWellAnchored(bestOrphan);
        const bool assumeWellAnchored = leanUpdateMappingScore(x, orphanF, orphanWellAnchored, bestOrphanIndex, orphanListIdx,
                                                               0 == u32(orphanF.editDistance) + u32(shadowF.editDistance) || shadowWellAnchored);
        t.properPair = TLS_NOMINAL == leanCheckModel(*x.tls, orphanF, shadowF);
        if (assumeWellAnchored)
        {
            const double shadowRog = leanRogRead(x, bestShadowIndex);
            const double otherShadowsProbability = (totalShadowProbability - exp(bestShadowC.logProbability)) + shadowRog;
            shadowF.alignmentScore = leanMapq(x, otherShadowsProbability / (totalShadowProbability + shadowRog));
            const double orphanRog = leanRogRead(x, bestOrphanIndex);
            const double otherOrphansProbability = (totalOrphanProbability - exp(bestOrphan.logProbability)) + orphanRog;
            orphanF.alignmentScore = leanMapq(x, otherOrphansProbability / (totalOrphanProbability + orphanRog));
            const double otherPairsProbability = (bo.total - exp(bo.bestLp)) + x.rog;
            t.alignmentScore = leanMapq(x, otherPairsProbability / (bo.total + x.rog));
            if ((!orphanF.alignmentScore || !orphanWellAnchored) && (!shadowF.alignmentScore || !shadowWellAnchored))
            { leanClampDodgy(t.alignmentScore); leanClampDodgy(shadowF.alignmentScore); leanClampDodgy(orphanF.alignmentScore); }
        }
        else ret = leanFlagDodgy(x, orphanF, shadowF, t);
        leanPut(t, bestOrphanIndex, orphanF); leanPut(t, bestShadowIndex, shadowF);
    }
    else if (known.resolved) ret = leanFlagDodgy(x, t.f0, t.f1, t);
    else
    {
        const Cand &c1 = x.l0[bd0], &c2 = x.l1[bd1];
        leanLoad(t.f0, c1, x.pool); leanLoad(t.f1, c2, x.pool);
        t.alignmentScore = 0; t.properPair = false;
        const bool w1 = candWellAnchored(c1), w2 = candWellAnchored(c2);
        const bool a1 = leanUpdateMappingScore(x, t.f0, w1, 0, bd0, 0 == t.f0.editDistance);
        const bool a2 = leanUpdateMappingScore(x, t.f1, w2, 1, bd1, 0 == t.f1.editDistance);
        if (!a1 && !a2) ret = leanFlagDodgy(x, t.f0, t.f1, t);
        else
        {
            if (!w1) leanClampDodgy(t.f0.alignmentScore);
            if (!w2) leanClampDodgy(t.f1.alignmentScore);
        }
    }
    return ret;
}

// BamTemplate::filterLowQualityFragments (BamTemplate.cpp:47-72)
ISAAC_HD bool leanFilterLowQuality(LeanTemplate &t, u32 mapqThreshold)
{
    bool ret = false; u32 alignmentScore = 0;
    if (mapqThreshold > t.f0.alignmentScore)
    {
        t.f0.cigarLength = 0; t.f0.alignmentScore = 0;
        if (2 == t.n) { t.f0.position = t.f1.position; t.f0.contigId = t.f1.contigId; }     // one read: the mate is the fragment itself
    }
    else if (leanAligned(t.f0)) ret = true;
    alignmentScore += t.f0.alignmentScore;
    if (2 == t.n)
    {
        if (mapqThreshold > t.f1.alignmentScore) { t.f1.cigarLength = 0; t.f1.alignmentScore = 0; t.f1.position = t.f0.position; t.f1.contigId = t.f0.contigId; }
        else if (leanAligned(t.f1)) ret = true;
        alignmentScore += t.f1.alignmentScore;
    }
    t.alignmentScore = alignmentScore;
    return ret;
}

// TemplateBuilder::buildTemplate (TemplateBuilder.cpp:97-174)
ISAAC_HD bool leanBuildTemplate(LeanCtx &x, const LeanRescue &rs, LeanTemplate &t, u32 &overflow)
{
    t.n = x.P->nReads; t.alignmentScore = 0; t.properPair = false;
    leanInit(t.f0, 0); leanInit(t.f1, 1);
    bool ret;
    if (2 == x.P->nReads)
    {
        if (x.n0 && x.n1)
        {
            LeanBestPair bc;
            leanLocateBestPair<true>(x, bc);
            bool paired = false; u32 i0 = 0, i1 = 0;
            if (bc.resolved)
            {
                leanChosenPair(x, bc, i0, i1);
                // buildPairedEndTemplate (TemplateBuilder.cpp:398-465)
                const Cand &c1 = x.l0[i0], &c2 = x.l1[i1];
                leanLoad(t.f0, c1, x.pool); leanLoad(t.f1, c2, x.pool);
                const bool w1 = candWellAnchored(c1), w2 = candWellAnchored(c2);
                const bool r1WellAnchored = leanUpdateMappingScore(x, t.f0, w1, 0, i0, w2);
                const bool r2WellAnchored = leanUpdateMappingScore(x, t.f1, w2, 1, i1, w1);
                t.properPair = TLS_NOMINAL == leanCheckModel(*x.tls, t.f0, t.f1);
                if (r1WellAnchored || r2WellAnchored)
                {
                    const double otherPairsProbability = (bc.total - exp(bc.bestLp)) + x.rog;
                    t.alignmentScore = leanMapq(x, otherPairsProbability / (bc.total + x.rog));
                    paired = r1WellAnchored && r2WellAnchored && !c1.repeatSeedsCount && !c2.repeatSeedsCount;
                }
                else t.alignmentScore = 0xffffffffu;
            }
            if (!bc.resolved || !paired || bc.editDistance) ret = leanDisjoinedTemplate(x, rs, t, bc, i0, i1, overflow);
            else ret = true;
        }
        else if (x.n0 || x.n1) ret = leanRescueShadow(x, rs, t, overflow);
        else ret = false;
    }
    else
    {   // pickBestFragment (TemplateBuilder.cpp:1035-1058)
        if (!x.n0) ret = false;
        else
        {
            const u32 best = leanBestFragment(x, 0);
            const Cand &c = x.l0[best];
            leanLoad(t.f0, c, x.pool);
            ret = true;
            if (!leanUpdateMappingScore(x, t.f0, candWellAnchored(c), 0, best, false))
            {
                if (-1 == x.P->dodgyAlignmentScore) { leanSetNoMatch(t.f0); t.alignmentScore = 0xffffffffu; ret = false; }
                else { t.f0.alignmentScore = 0xffffffffu; t.alignmentScore = 0xffffffffu; }
            }
        }
    }
    if (overflow) return false;
    if (ret && 0xffffffffu != t.alignmentScore)
    {
        if (!t.properPair) ret = leanFilterLowQuality(t, x.P->mapqThreshold);
        else if (x.P->mapqThreshold > t.alignmentScore) { leanFilterLowQuality(t, 0xffffffffu); ret = false; }
    }
    return ret;
}

// ---- the end clippers.  The CIGAR of a template end stays where it is (the cluster's pool, a rescued shadow's three words, a gapped result); the clippers
// look at its first two and last two words (LeanFrag::w0, w1, wl, wp) and note what they take from its first and last ALIGN operation
// (frontClip, backClip); leanWriteCigar writes it once, clipped, into the record's output slot.
ISAAC_HD void leanLoadEdges(LeanFrag &f)
{
    const u32 n = f.cigarLength;
    f.w0 = n ? f.cigar[0] : 0u; f.w1 = n > 1 ? f.cigar[1] : 0u; f.wl = n ? f.cigar[n - 1] : 0u; f.wp = n > 1 ? f.cigar[n - 2] : 0u;
    f.frontClip = 0; f.backClip = 0;
}
ISAAC_HD bool leanSrcFrontClip(const LeanFrag &f) { return f.cigarLength && OP_SOFT_CLIP == cigarCode(f.w0); }
ISAAC_HD bool leanSrcBackClip(const LeanFrag &f) { return f.cigarLength && OP_SOFT_CLIP == cigarCode(f.wl); }
// the CIGAR as it stands: its length, its soft clips, and the operations next to them
ISAAC_HD u32 leanCigarLengthNow(const LeanFrag &f) { return u32(f.cigarLength) + ((f.frontClip && !leanSrcFrontClip(f)) ? 1u : 0u) + ((f.backClip && !leanSrcBackClip(f)) ? 1u : 0u); }
ISAAC_HD bool leanHasFrontClip(const LeanFrag &f) { return leanSrcFrontClip(f) || f.frontClip; }
ISAAC_HD bool leanHasBackClip(const LeanFrag &f) { return leanSrcBackClip(f) || f.backClip; }
ISAAC_HD u32 leanFrontClipBases(const LeanFrag &f) { return (leanSrcFrontClip(f) ? cigarLen(f.w0) : 0u) + f.frontClip; }
ISAAC_HD u32 leanBackClipBases(const LeanFrag &f) { return (leanSrcBackClip(f) ? cigarLen(f.wl) : 0u) + f.backClip; }
// index (in the CIGAR where it lies) of the operation behind the leading soft clip / before the trailing one
ISAAC_HD i32 leanFirstInner(const LeanFrag &f) { return leanSrcFrontClip(f) ? 1 : 0; }
ISAAC_HD i32 leanLastInner(const LeanFrag &f) { return i32(f.cigarLength) - (leanSrcBackClip(f) ? 2 : 1); }
// those operations as they stand (0: there is none): an ALIGN operation less what the clippers have taken from it
ISAAC_HD u32 leanFirstInnerOp(const LeanFrag &f)
{
    const i32 i = leanFirstInner(f), n = i32(f.cigarLength);
    if (i >= n) return 0;
    const u32 w = 0 == i ? f.w0 : f.w1;
    if (OP_ALIGN != cigarCode(w)) return w;
    return cigarOp(cigarLen(w) - f.frontClip - (i == leanLastInner(f) ? u32(f.backClip) : 0u), OP_ALIGN);
}
ISAAC_HD u32 leanLastInnerOp(const LeanFrag &f)
{
    const i32 i = leanLastInner(f), n = i32(f.cigarLength);
    if (i < 0) return 0;
    const u32 w = n - 1 == i ? f.wl : f.wp;
    if (OP_ALIGN != cigarCode(w)) return w;
    return cigarOp(cigarLen(w) - f.backClip - (i == leanFirstInner(f) ? u32(f.frontClip) : 0u), OP_ALIGN);
}
// the CIGAR as it stands into the record's slot (room for the source's words + 2); returns its length
ISAAC_HD u32 leanWriteCigar(const LeanFrag &f, u32 *out)
{
    const u32 n = f.cigarLength;
    if (!f.frontClip && !f.backClip) { for (u32 k = 0; k < n; ++k) out[k] = f.cigar[k]; return n; }
    const i32 iF = leanFirstInner(f), iL = leanLastInner(f);
    u32 k = 0;
    if (leanHasFrontClip(f)) out[k++] = cigarOp(leanFrontClipBases(f), OP_SOFT_CLIP);
    for (i32 i = iF; i <= iL; ++i)
    {
        u32 w = i == iF ? leanFirstInnerOp(f) : i == iL ? leanLastInnerOp(f) : f.cigar[i];
        out[k++] = w;
    }
    if (leanHasBackClip(f)) out[k++] = cigarOp(leanBackClipBases(f), OP_SOFT_CLIP);
    return k;
}

// clipMismatches<5> (Alignment.hh:55-88; template.h: clipMismatches) with the first eight bases of either side fetched at once: nearly every end
// is decided by its first five bases, and the loop over single bytes is a memory round trip per base.  A LeanScan is such a scan asked for (leanScanAsk:
// the two loads are issued) and not yet looked at (leanScanRun): the four scans of a pair -- either end of either read -- are asked for together, one
// round trip for all of them.  refIdx: position on contig contigId.
struct LeanScan { u64 readBytes, refBytes; i64 seqIdx, refIdx, b0, blo, a0, alo; i32 dir, bdir; bool reverse, fetched; u32 contigId; };
ISAAC_HD void leanScanAsk(LeanScan &w, const LeanCtx &x, const ReadView &read, bool reverse, i64 seqIdx, u32 contigId, i64 refIdx, i32 dir)
{
    w.seqIdx = seqIdx; w.refIdx = refIdx; w.dir = dir; w.reverse = reverse; w.contigId = contigId; w.fetched = false; w.readBytes = 0; w.refBytes = 0;
    w.b0 = w.blo = w.a0 = w.alo = 0; w.bdir = dir;
    const i64 total = i64(x.R->totalBases), L = i64(read.length);
    if (L < 8 || total < 8 || MAX_CONTIG_ID == contigId) return;
    w.b0 = reverse ? L - 1 - seqIdx : seqIdx;                        // BCL byte of the first base looked at, and which way the bytes go
    w.bdir = reverse ? -dir : dir;
    i64 blo = w.bdir > 0 ? w.b0 : w.b0 - 7; blo = blo < 0 ? 0 : blo > L - 8 ? L - 8 : blo;
    w.a0 = i64(x.R->contigOffset[contigId]) + refIdx;
    i64 alo = dir > 0 ? w.a0 : w.a0 - 7; alo = alo < 0 ? 0 : alo > total - 8 ? total - 8 : alo;
    w.blo = blo; w.alo = alo;
    memcpy(&w.readBytes, read.bcl + blo, 8); memcpy(&w.refBytes, x.R->bases + alo, 8);
    w.fetched = true;
}
ISAAC_HD void leanScanRun(const LeanScan &w, const LeanCtx &x, const ReadView &read, i64 seqCount, i64 refCount, u32 &clipped, u32 &editAdj)
{
    if (w.fetched)
    {
        const u32 MIN = 5;
        u32 matchesInARow = 0, edMismatches = 0, edUnclipped = 0, k = 0;
        bool inWindow = true;
        for (; k < 8 && i64(k) < seqCount && i64(k) < refCount && MIN > matchesInARow; ++k)
        {
            const i64 bi = w.b0 + i64(w.bdir) * i64(k) - w.blo, ai = w.a0 + i64(w.dir) * i64(k) - w.alo;
            if (bi < 0 || bi > 7 || ai < 0 || ai > 7) { inWindow = false; break; }
            const u8 b = u8(w.readBytes >> (8 * u32(bi))); const char r = char(w.refBytes >> (8 * u32(ai)));
            const char s = !(b & 0xfc) ? 'n' : char(0x54474341u >> (8 * (w.reverse ? (~u32(b)) & 3u : u32(b) & 3u)));
            if (isMatch(s, r)) { ++matchesInARow; edUnclipped += (s != r); }
            else { matchesInARow = 0; edUnclipped = 0; }
            edMismatches += (s != r);
        }
        if (inWindow && !(i64(k) < seqCount && i64(k) < refCount && MIN > matchesInARow))
        {   // the loop has run to its end
            if (MIN == matchesInARow) { clipped = k - matchesInARow; editAdj = edMismatches - edUnclipped; }
            else { clipped = 0; editAdj = 0; }
            return;
        }
    }
    clipMismatches(read, w.reverse, w.seqIdx, seqCount, x.R->bases + x.R->contigOffset[w.contigId], w.refIdx, refCount, w.dir, clipped, editAdj);
}

// SemialignedEndsClipper::clipLeftSide / clipRightSide (SemialignedEndsClipper.cpp:31-156).  Where either scan begins does not depend on the other's
// outcome (a clip at the left moves position and observed length together): leanSemialignedAsk issues the loads of both before either is looked at.
struct LeanEndScans { LeanScan left, right; };
ISAAC_HD void leanSemialignedAsk(LeanEndScans &w, const LeanCtx &x, const ReadView &read, const LeanFrag &f)
{
    leanScanAsk(w.left, x, read, f.reverse, i64(leanHasFrontClip(f) ? leanFrontClipBases(f) : 0u), f.contigId, f.position, +1);
    leanScanAsk(w.right, x, read, f.reverse, i64(read.length) - 1 - i64(leanHasBackClip(f) ? leanBackClipBases(f) : 0u), f.contigId, f.position + i64(leanObservedLength(f)) - 1, -1);
}
ISAAC_HD bool leanSemialignedLeft(const LeanCtx &x, const ReadView &read, LeanFrag &f, const LeanScan &asked)
{
    i64 seqBegin = 0;
    if (leanHasFrontClip(f))
    {
        if (2 > leanCigarLengthNow(f)) return false;
        seqBegin += leanFrontClipBases(f);
    }
    const u32 op = leanFirstInnerOp(f);
    if (!op || OP_ALIGN != cigarCode(op)) return false;
    const i64 refSize = i64(contigLength(*x.R, f.contigId));
    u32 clipped, editAdj;
    if (asked.seqIdx == seqBegin && asked.refIdx == f.position) leanScanRun(asked, x, read, cigarLen(op), refSize - f.position, clipped, editAdj);
    else clipMismatches(read, f.reverse, seqBegin, cigarLen(op), x.R->bases + x.R->contigOffset[f.contigId], f.position, refSize - f.position, +1, clipped, editAdj);
    if (!clipped) return false;
    f.observedLength -= clipped; f.position += clipped; f.editDistance = u16(f.editDistance - editAdj);
    f.frontClip = u16(f.frontClip + clipped);
    return true;
}
ISAAC_HD bool leanSemialignedRight(const LeanCtx &x, const ReadView &read, LeanFrag &f, const LeanScan &asked)
{
    i64 seqR = i64(read.length) - 1;
    if (leanHasBackClip(f))
    {
        if (2 > leanCigarLengthNow(f)) return false;
        seqR -= leanBackClipBases(f);
    }
    const u32 op = leanLastInnerOp(f);
    if (!op || OP_ALIGN != cigarCode(op)) return false;
    const i64 refLast = f.position + i64(leanObservedLength(f)) - 1;
    u32 clipped, editAdj;
    if (asked.seqIdx == seqR && asked.refIdx == refLast) leanScanRun(asked, x, read, cigarLen(op), refLast + 1, clipped, editAdj);
    else clipMismatches(read, f.reverse, seqR, cigarLen(op), x.R->bases + x.R->contigOffset[f.contigId], refLast, refLast + 1, -1, clipped, editAdj);
    if (!clipped) return false;
    f.observedLength -= clipped; f.editDistance = u16(f.editDistance - editAdj);
    f.backClip = u16(f.backClip + clipped);
    return true;
}

// OverlappingEndsClipper::clip (OverlappingEndsClipper.cpp:46-183); left / right: the template's ends by position (copies: the caller
// writes the one that changed back, so that nothing here is reached through a pointer chosen at run time).
// Returns 0: nothing clipped, 1: `right` clipped at its start, 2: `left` clipped at its end.
ISAAC_HD u32 leanOverlappingClipOrdered(const LeanCtx &x, const ReadView &leftRead, const ReadView &rightRead, LeanFrag &left, LeanFrag &right)
{
    if (left.reverse) return 0;
    const i64 overlapLength = left.position + i64(leanObservedLength(left)) - right.position;
    if (0 >= overlapLength) return 0;
    u32 leftEndOffset = leftRead.length;
    if (leanHasBackClip(left))
    {
        if (leanCigarLengthNow(left) < 2) return 0;
        leftEndOffset -= leanBackClipBases(left);
    }
    const u32 leftLastOp = leanLastInnerOp(left);
    if (!leftLastOp || OP_ALIGN != cigarCode(leftLastOp)) return 0;
    if (overlapLength >= i64(cigarLen(leftLastOp))) return 0;
    u32 rightStartOffset = 0;
    if (leanHasFrontClip(right))
    {
        if (leanCigarLengthNow(right) < 2) return 0;
        rightStartOffset += leanFrontClipBases(right);
    }
    const u32 rightFirstOp = leanFirstInnerOp(right);
    if (!rightFirstOp || OP_ALIGN != cigarCode(rightFirstOp)) return 0;
    if (overlapLength >= i64(cigarLen(rightFirstOp))) return 0;
    i32 diff = 0;
    for (i64 i = 0; i < overlapLength; ++i)
        diff += i32(strandQuality(leftRead, false, u32(leftEndOffset - overlapLength + i))) - i32(strandQuality(rightRead, true, u32(rightStartOffset + i)));
    if (0 < diff)
    {
        const char *reference = x.R->bases + x.R->contigOffset[right.contigId] + right.position;
        u32 ed = 0; for (i64 i = 0; i < overlapLength; ++i) ed += (strandBase(rightRead, true, u32(rightStartOffset + i)) != reference[i]);
        right.frontClip = u16(right.frontClip + u32(overlapLength));
        right.position += overlapLength; if (right.reverse) right.highClipped += u16(overlapLength); else right.lowClipped += u16(overlapLength);   // incrementClipLeft
        right.observedLength -= u32(overlapLength);
        right.editDistance = u16(right.editDistance - ed);
        return 1;
    }
    const char *reference = x.R->bases + x.R->contigOffset[left.contigId] + left.position + i64(leanObservedLength(left)) - overlapLength;
    u32 ed = 0; for (i64 i = 0; i < overlapLength; ++i) ed += (strandBase(leftRead, false, u32(leftEndOffset - overlapLength + i)) != reference[i]);
    left.backClip = u16(left.backClip + u32(overlapLength));
    if (left.reverse) left.lowClipped += u16(overlapLength); else left.highClipped += u16(overlapLength);                                       // incrementClipRight
    left.observedLength -= u32(overlapLength);
    left.editDistance = u16(left.editDistance - ed);
    return 2;
}

// io::FragmentHeader of one end (template.h: makeFragmentRecord)
ISAAC_HD void leanRecord(const LeanCtx &x, const LeanTemplate &t, const LeanFrag &f, const LeanFrag &mate, u32 readLength, u32 tile, FragmentRecord &r)
{
    const u16 DODGY = 0xffff;
    if (2 == t.n)
    {
        i32 tlen = 0;
        if (leanAligned(f) && leanAligned(mate))
        {
            const u64 fb = leanFStrandPos(f), fe = refpos(f.contigId, u64(f.position + i64(f.observedLength)));
            const u64 mb = leanFStrandPos(mate), me = refpos(mate.contigId, u64(mate.position + i64(mate.observedLength)));
            const u64 distance = refposLocation(imax(fe, me)) - refposLocation(imin(fb, mb));
            const bool firstRead = 0 == f.readIndex;
            tlen = i32(fb < mb ? i64(distance) : (mb < fb || !firstRead) ? -i64(distance) : i64(distance));
        }
        r.bamTlen = tlen;
        r.fStrandPosition = leanAligned(f) ? leanFStrandPos(f) : leanFStrandPos(mate);
        r.templateAlignmentScore = u16(t.properPair ? t.alignmentScore : f.alignmentScore);
        r.mateFStrandPosition = leanAligned(mate) ? leanFStrandPos(mate) : leanFStrandPos(f);
        r.flags = 1u | (u32(!leanAligned(f)) << 1) | (u32(!leanAligned(mate)) << 2) | (u32(f.reverse) << 3) | (u32(mate.reverse) << 4) |
                  (u32(0 == f.readIndex) << 5) | (u32(1 == f.readIndex) << 6) | (u32(t.properPair) << 8);
    }
    else
    {
        r.bamTlen = 0; r.fStrandPosition = leanFStrandPos(f); r.templateAlignmentScore = u16(f.alignmentScore); r.mateFStrandPosition = REFPOS_NOMATCH;
        r.flags = (u32(!leanAligned(f)) << 1) | (1u << 2) | (u32(f.reverse) << 3) | (1u << 5) | (1u << 6);
    }
    r.observedLength = leanObservedLength(f);
    r.lowClipped = f.lowClipped; r.highClipped = f.highClipped; r.alignmentScore = u16(f.alignmentScore);
    r.readLength = u16(readLength); r.cigarLength = f.cigarLength; r.gapCount = f.gapCount; r.editDistance = f.editDistance;
    r.tile = tile; r.clusterId = x.clusterId; r.reserved = 0; r.cigarOffset = 0;
    const u32 forced = u32(x.P->dodgyAlignmentScore) & 0xff;
    if (r.flags & (1u << 8)) r.mapq = (DODGY == r.templateAlignmentScore) ? forced : imin<u32>(60u, imax(r.alignmentScore, r.templateAlignmentScore));
    else r.mapq = (DODGY == r.alignmentScore) ? forced : imin<u32>(60u, r.alignmentScore);
}

static const u32 LEAN_OUT_CIGAR_CAP = 40;   // == OUT_CIGAR_CAP (cluster_ops.h)

// MatchSelector::processMatchList for one cluster on precomputed rescue outcomes (cluster_ops.h: clusterSelect in RESCUE_PRECOMPUTED mode):
// template, clippers, records.  Returns false when the cluster needs the general form instead (--scatter-repeats picks a rescued placement
// beyond the LEAN_TIES kept, or a CIGAR is too long to be clipped in its output slot); the records are then not valid.
// l0, l1: the cluster's two candidate lists -- where they lie in the pool, or a copy of them (k_select: in LDS); their CIGAR offsets are
// relative to the cluster's words of the arena either way
ISAAC_HD bool leanSelectCluster(const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, double logMismatchQ40, const u8 *bcl, u32 cluster, u32 tile,
                                const ClusterMeta &meta, const Cand *l0, const Cand *l1, const u32 *cigarArena, const LeanRescue &rs, FragmentRecord *records, u32 *cigars, u32 &mapqNearInteger)
{
    LeanCtx x;
    x.P = &P; x.R = &R; x.tls = &tls;
    x.l0 = l0; x.l1 = l1; x.n0 = meta.nCands[0]; x.n1 = meta.nCands[1];
    x.pool = cigarArena + 3 * u64(meta.first);
    x.rogRead0 = rog.read[0]; x.rogRead1 = rog.read[1]; x.rog = rog.pair; x.logMismatchQ40 = logMismatchQ40;
    x.clusterId = cluster; x.mapqNearInteger = 0;
    LeanTemplate t;
    u32 overflow = 0;
    bool store;
#if defined(ISAAC_TIMING_SELECT_NO_TEMPLATE)
    if (meta.built && cluster == 0xfffffff0u) store = leanBuildTemplate(x, rs, t, overflow) || P.keepUnaligned;
    else if (meta.built) { t.n = P.nReads; t.alignmentScore = 0; t.properPair = false; leanInit(t.f0, 0); leanInit(t.f1, 1); if (x.n0) leanLoad(t.f0, x.l0[0], x.pool); if (x.n1) leanLoad(t.f1, x.l1[0], x.pool); store = true; }
#else
    if (meta.built) store = leanBuildTemplate(x, rs, t, overflow) || P.keepUnaligned;
#endif
    else { t.n = P.nReads; t.alignmentScore = 0; t.properPair = false; leanInit(t.f0, 0); leanInit(t.f1, 1); store = 0 != P.keepUnaligned; }
    if (overflow) return false;
    if (!store) { t.n = P.nReads; t.alignmentScore = 0; t.properPair = false; leanInit(t.f0, 0); leanInit(t.f1, 1); }
    if (u32(t.f0.cigarLength) + 2 > LEAN_OUT_CIGAR_CAP || u32(t.f1.cigarLength) + 2 > LEAN_OUT_CIGAR_CAP) return false;
    u32 *cig0 = cigars + (u64(cluster) * P.nReads) * LEAN_OUT_CIGAR_CAP, *cig1 = cig0 + LEAN_OUT_CIGAR_CAP;
    leanLoadEdges(t.f0); leanLoadEdges(t.f1);
    const u8 *clusterBcl = bcl + u64(cluster) * P.clusterLength;
    ReadView read0, read1;
    read0.bcl = clusterBcl + P.readOffset[0]; read0.length = P.readLength[0]; read0.firstCycle = P.firstCycle[0]; read0.endCyclesMasked = meta.endCyclesMasked[0];
    read1.bcl = clusterBcl + P.readOffset[1]; read1.length = 1 < P.nReads ? P.readLength[1] : 0; read1.firstCycle = P.firstCycle[1]; read1.endCyclesMasked = meta.endCyclesMasked[1];
    // -DISAAC_TIMING_SELECT_NO_CLIP / _NO_TEMPLATE: builds that leave a section out (wrong results) to time the others
#if defined(ISAAC_TIMING_SELECT_NO_CLIP)
    if (store && meta.built && cluster == 0xfffffff0u)
#else
    if (store && meta.built)
#endif
    {
        if (P.clipSemialigned)
        {   // SemialignedEndsClipper::clip (SemialignedEndsClipper.cpp:161-205); the four scans' first bytes are asked for before any is looked at
            LeanEndScans scans0, scans1;
            if (leanAligned(t.f0)) leanSemialignedAsk(scans0, x, read0, t.f0); else { scans0.left.fetched = scans0.right.fetched = false; scans0.left.seqIdx = scans0.right.seqIdx = -1; scans0.left.refIdx = scans0.right.refIdx = 0; }
            if (2 == t.n && leanAligned(t.f1)) leanSemialignedAsk(scans1, x, read1, t.f1); else { scans1.left.fetched = scans1.right.fetched = false; scans1.left.seqIdx = scans1.right.seqIdx = -1; scans1.left.refIdx = scans1.right.refIdx = 0; }
            bool stop = false;
            if (leanAligned(t.f0))
            {
                bool changed = leanSemialignedLeft(x, read0, t.f0, scans0.left);
                if (leanSemialignedRight(x, read0, t.f0, scans0.right)) changed = true;
                if (changed && 2 == t.n && !leanAligned(t.f1)) { t.f1.position = t.f0.position; stop = true; }
            }
            if (!stop && 2 == t.n && leanAligned(t.f1))
            {
                bool changed = leanSemialignedLeft(x, read1, t.f1, scans1.left);
                if (leanSemialignedRight(x, read1, t.f1, scans1.right)) changed = true;
                if (changed && !leanAligned(t.f0)) t.f0.position = t.f1.position;
            }
        }
        if (P.clipOverlapping && 2 == t.n && leanAligned(t.f0) && leanAligned(t.f1) && !t.f0.gapCount && !t.f1.gapCount && t.f0.reverse != t.f1.reverse)
        {
            // left: the end that starts first; when both start together OverlappingEndsClipper.cpp:68-69 makes r2 both left and right
            const bool leftIs0 = t.f0.position < t.f1.position, rightIs0 = !(t.f0.position <= t.f1.position);
            LeanFrag left = leanPick(leftIs0, t.f0, t.f1), right = leanPick(rightIs0, t.f0, t.f1);
            const u32 changed = leanOverlappingClipOrdered(x, leanPickRead(leftIs0, read0, read1), leanPickRead(rightIs0, read0, read1), left, right);
            if (1 == changed) leanPut(t, rightIs0 ? 0 : 1, right);
            else if (2 == changed) leanPut(t, leftIs0 ? 0 : 1, left);
        }
    }
    // the CIGARs, clipped, into their output slots (an unaligned end has none)
    if (leanAligned(t.f0)) t.f0.cigarLength = u16(leanWriteCigar(t.f0, cig0));
    if (2 == t.n && leanAligned(t.f1)) t.f1.cigarLength = u16(leanWriteCigar(t.f1, cig1));
    mapqNearInteger = x.mapqNearInteger;
    const u32 templateScore = (t.alignmentScore >= 0xffffu ? 0xffffu : t.alignmentScore) << 16;
    const u32 reserved = ((meta.flags & CLUSTER_OVERFLOW) ? u32(RECORD_FRAGMENT_OVERFLOW) : 0u) | (store ? 0u : u32(RECORD_NOT_STORED)) | (x.mapqNearInteger ? u32(RECORD_MAPQ_NEAR_INTEGER) : 0u) | templateScore;
    {
        FragmentRecord &r = records[u64(cluster) * P.nReads];
        leanRecord(x, t, t.f0, t.f1, read0.length, tile, r);
        r.cigarOffset = u32((u64(cluster) * P.nReads) * LEAN_OUT_CIGAR_CAP); r.reserved = reserved;
    }
    if (2 == P.nReads)
    {
        FragmentRecord &r = records[u64(cluster) * 2 + 1];
        leanRecord(x, t, t.f1, t.f0, read1.length, tile, r);
        r.cigarOffset = u32((u64(cluster) * 2 + 1) * LEAN_OUT_CIGAR_CAP); r.reserved = reserved;
    }
    return true;
}
// the same on the lists where they lie in the chunk's pool
ISAAC_HD bool leanSelectCluster(const DevParams &P, const DevReference &R, const DevTls &tls, const RogCorrection &rog, double logMismatchQ40, const u8 *bcl, u32 cluster, u32 tile,
                                const ClusterMeta &meta, const Cand *candPool, const u32 *cigarArena, const LeanRescue &rs, FragmentRecord *records, u32 *cigars, u32 &mapqNearInteger)
{
    return leanSelectCluster(P, R, tls, rog, logMismatchQ40, bcl, cluster, tile, meta, candPool + meta.first, candPool + meta.first + meta.second, cigarArena, rs, records, cigars, mapqNearInteger);
}

} // namespace isaac
