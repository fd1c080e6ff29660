// Thread-serial template stage: template-length model checks, mate rescue (ShadowAligner), pair / orphan selection and
// alignment scores (TemplateBuilder), the semialigned / overlapping end clippers and the io::FragmentHeader record.
//
// Behaviour follows (paths relative to /root/reference/src/c++):
//   include/alignment/TemplateLengthStatistics.hh:104-239, lib/alignment/TemplateLengthStatistics.cpp:67-239,
//   lib/alignment/ShadowAligner.cpp:53-291, lib/alignment/TemplateBuilder.cpp:52-1086, include/alignment/TemplateBuilder.hh:166-306,
//   include/alignment/RestOfGenomeCorrection.hh:44-88, lib/alignment/BamTemplate.cpp:47-72,
//   lib/alignment/matchSelector/SemialignedEndsClipper.cpp:31-205, OverlappingEndsClipper.cpp:46-183,
//   include/io/Fragment.hh:101-246, include/build/FragmentAccessorBamAdapter.hh:250-265, lib/alignment/MatchSelector.cpp:258-368
#pragma once
#include "aligner.h"
#if !defined(__HIPCC__)
#include <cmath>
#endif
#ifdef ISAAC_DEBUG_MAPQ
#include <cstdio>
#endif

namespace isaac
{

struct DevTls { u32 min, max, median, lowStdDev, highStdDev; i32 bestModel[2]; u32 stable, mateMin, mateMax; };
static const u32 TEMPLATE_LENGTH_THRESHOLD = 50000;

ISAAC_HD i32 tlsAlignmentModel(const Cand &f1, const Cand &f2)
{
    if (f1.contigId == f2.contigId) return i32(((f1.position <= f2.position) ? 0 : 4) | (f1.reverse ? 2 : 0) | (f2.reverse ? 1 : 0));
    return 8;
}
ISAAC_HD u64 tlsGetLength(const Cand &f1, const Cand &f2)
{
    if (f1.position < f2.position) return u64(imax<i64>(f2.position + i64(candObservedLength(f2)) - f1.position, i64(candObservedLength(f1))));
    return u64(imax<i64>(f1.position + i64(candObservedLength(f1)) - f2.position, i64(candObservedLength(f2))));
}
enum { TLS_OVERSIZED = 0, TLS_UNDERSIZED = 1, TLS_NOMINAL = 2, TLS_NOMATCH = 3 };
ISAAC_HD u32 tlsCheckModel(const DevTls &t, const Cand &f1, const Cand &f2)
{
    if (f1.contigId == f2.contigId)
    {
        const i32 model = tlsAlignmentModel(f1, f2);
        if (model == t.bestModel[0] || model == t.bestModel[1])
        {
            const u64 length = tlsGetLength(f1, f2);
            return (length > t.max) ? TLS_OVERSIZED : (length < t.min) ? TLS_UNDERSIZED : TLS_NOMINAL;
        }
    }
    return TLS_NOMATCH;
}
ISAAC_HD bool tlsMatchModel(const DevTls &t, const Cand &f1, const Cand &f2)
{
    const u64 length = tlsGetLength(f1, f2);
    const i32 model = tlsAlignmentModel(f1, f2);
    return (length <= u64(t.max + TEMPLATE_LENGTH_THRESHOLD)) && ((model == t.bestModel[0]) || (model == t.bestModel[1]));  // max_ + threshold is unsigned arithmetic
}
ISAAC_HD u32 tlsAlignmentClass(i32 m) { return (m < 4) ? u32(m) : ((~u32(m)) & 3); }
ISAAC_HD bool tlsIsCoherent(const DevTls &t) { return t.bestModel[0] != t.bestModel[1] && tlsAlignmentClass(t.bestModel[0]) == tlsAlignmentClass(t.bestModel[1]); }
ISAAC_HD bool tlsIsValidModel(const DevTls &t, bool reverse, u32 readIndex)
{ const u32 shift = (readIndex + 1) % 2; return (reverse == bool((t.bestModel[0] >> shift) & 1)) || (reverse == bool((t.bestModel[1] >> shift) & 1)); }
ISAAC_HD bool tlsFirstFragment(const DevTls &t, bool reverse, u32 readIndex)
{
    const u32 shift = (readIndex + 1) % 2;
    for (u32 i = 0; 2 > i; ++i) if (reverse == bool((t.bestModel[i] >> shift) & 1)) return u32((t.bestModel[i] >> 2) & 1) == readIndex;
    return false;
}
ISAAC_HD bool tlsMateOrientation(const DevTls &t, u32 readIndex, bool reverse)
{
    const u32 shift = (readIndex + 1) % 2;
    for (u32 i = 0; 2 > i; ++i) if (reverse == bool((t.bestModel[i] >> shift) & 1)) return (t.bestModel[i] >> readIndex) & 1;
    return (t.bestModel[0] >> readIndex) & 1;
}
// `position + mateMin_ - readLength` mixes long and unsigned operands: the unsigned ones are converted to long first
ISAAC_HD i64 tlsMateMinPosition(const DevTls &t, u32 readIndex, bool reverse, i64 position, const u32 *readLengths)
{
    if (!tlsIsValidModel(t, reverse, readIndex)) return position;
    if (tlsFirstFragment(t, reverse, readIndex)) return position + i64(t.mateMin) - i64(readLengths[(readIndex + 1) % 2]);
    return position - i64(t.mateMax) + i64(readLengths[readIndex]);
}
ISAAC_HD i64 tlsMateMaxPosition(const DevTls &t, u32 readIndex, bool reverse, i64 position, const u32 *readLengths)
{
    if (!tlsIsValidModel(t, reverse, readIndex)) return position;
    if (tlsFirstFragment(t, reverse, readIndex)) return position + i64(t.mateMax) - i64(readLengths[(readIndex + 1) % 2]);
    return position - i64(t.mateMin) + i64(readLengths[readIndex]);
}

// ------------------------------------------------------------------------------------------------------------------
// Work-list capacities are runtime values: the main pass runs with small per-thread lists (LIGHT); the rare clusters that
// overflow them are redone by a second launch with the reference's own limits (HEAVY: 10000 candidate positions and 1000
// tracked shadows, ShadowAligner.hh:91 / TemplateBuilder.hh:149).
struct TemplateCaps { u32 shadow, shadowCigar, pos, prob, pair, best, templateCigar, kmerTable, tflags; };   // kmerTable / tflags: only the serial rescue needs them
static const u32 KMER_TABLE = 1024;
ISAAC_HD TemplateCaps lightCaps() { TemplateCaps c; c.shadow = 48; c.shadowCigar = 512; c.pos = 384; c.prob = 256; c.pair = 256; c.best = 12; c.templateCigar = 768; c.kmerTable = KMER_TABLE; c.tflags = 3 * 512; return c; }
// the main pass (k_select, k_plan_rescue): rescue results and probability sums arrive precomputed (RESCUE_PRECOMPUTED), so a thread
// keeps only the tie lists of the best pairs, the clones of the best rescued shadows and the template's CIGAR buffer -- in
// private memory.  What does not fit goes to the wave-per-cluster pass with the reference's own limits.
#ifndef ISAAC_TINY_BEST
#define ISAAC_TINY_BEST 4           // equally good placements k_select keeps per read (a test build sets 1: most clusters then take the residual pass)
#endif
ISAAC_HD TemplateCaps tinyCaps() { TemplateCaps c; c.shadow = 0; c.shadowCigar = 0; c.pos = 0; c.prob = 0; c.pair = 0; c.best = ISAAC_TINY_BEST; c.templateCigar = 96; c.kmerTable = 0; c.tflags = 0; return c; }
ISAAC_HD TemplateCaps heavyCaps() { TemplateCaps c; c.shadow = 1000; c.shadowCigar = 16384; c.pos = 10000; c.prob = 32768; c.pair = 32768; c.best = 1000; c.templateCigar = 65536; c.kmerTable = KMER_TABLE; c.tflags = 3 * 512; return c; }
static const u32 TRACKED_REPEATS_MAX_ONE_READ = 1000;
static const u32 SKIP_ORPHAN_EDIT_DISTANCE = 3, DODGY_BUT_CLEAN_ALIGNMENT_SCORE = 10;

struct ShadowProb { u64 pos; double logProbability; i64 observedLength; };
struct PairProb { ShadowProb r1, r2; };
ISAAC_HD u64 candFStrandPos(const Cand &c) { return candNoMatch(c) ? REFPOS_NOMATCH : refpos(c.contigId, u64(c.position)); }
ISAAC_HD u64 candRStrandPos(const Cand &c) { return candNoMatch(c) ? REFPOS_NOMATCH : refpos(c.contigId, u64(imax<i64>(c.position + i64(c.observedLength), 1) - 1)); }
ISAAC_HD ShadowProb makeShadowProb(const Cand &s)
{ ShadowProb p; p.pos = (candFStrandPos(s) & ~u64(1)) | u64(s.reverse ? 1 : 0); p.logProbability = s.logProbability; p.observedLength = i64(candObservedLength(s)); return p; }
ISAAC_HD bool shadowProbLess(const ShadowProb &a, const ShadowProb &b)
{ return a.pos < b.pos || (a.pos == b.pos && (lpLess(a.logProbability, b.logProbability) || (lpEquals(a.logProbability, b.logProbability) && a.observedLength < b.observedLength))); }
ISAAC_HD bool shadowProbEqual(const ShadowProb &a, const ShadowProb &b) { return a.pos == b.pos && lpEquals(a.logProbability, b.logProbability) && a.observedLength == b.observedLength; }
ISAAC_HD double pairLp(const PairProb &p) { return p.r1.logProbability + p.r2.logProbability; }
ISAAC_HD bool pairProbLess(const PairProb &a, const PairProb &b)
{
    return a.r1.pos < b.r1.pos || (a.r1.pos == b.r1.pos && (a.r2.pos < b.r2.pos || (a.r2.pos == b.r2.pos &&
           (lpLess(pairLp(b), pairLp(a)) || (lpEquals(pairLp(a), pairLp(b)) &&
           (a.r1.observedLength < b.r1.observedLength || (a.r1.observedLength == b.r1.observedLength && a.r2.observedLength < b.r2.observedLength)))))));
}
ISAAC_HD bool pairProbEqual(const PairProb &a, const PairProb &b)
{ return a.r1.pos == b.r1.pos && a.r2.pos == b.r2.pos && lpEquals(pairLp(a), pairLp(b)) && a.r1.observedLength == b.r1.observedLength && a.r2.observedLength == b.r2.observedLength; }
struct ShadowProbIdxLess { const ShadowProb *v; ISAAC_HD bool operator()(u16 a, u16 b) const { return shadowProbLess(v[a], v[b]); } };
struct PairProbIdxLess { const PairProb *v; ISAAC_HD bool operator()(u16 a, u16 b) const { return pairProbLess(v[a], v[b]); } };
// total orders refining shadowProbLess / pairProbLess (exact comparisons, index as the last key): see sumUnique*Probabilities
struct ShadowProbTotalLess
{
    const ShadowProb *v;
    ISAAC_HD bool operator()(u16 a, u16 b) const
    {
        const ShadowProb &x = v[a], &y = v[b];
        if (x.pos != y.pos) return x.pos < y.pos;
        if (x.logProbability != y.logProbability) return x.logProbability < y.logProbability;
        if (x.observedLength != y.observedLength) return x.observedLength < y.observedLength;
        return a < b;
    }
};
struct PairProbTotalLess
{
    const PairProb *v;
    ISAAC_HD bool operator()(u16 a, u16 b) const
    {
        const PairProb &x = v[a], &y = v[b];
        if (x.r1.pos != y.r1.pos) return x.r1.pos < y.r1.pos;
        if (x.r2.pos != y.r2.pos) return x.r2.pos < y.r2.pos;
        const double lx = pairLp(x), ly = pairLp(y);
        if (lx != ly) return ly < lx;                                   // higher probability first
        if (x.r1.observedLength != y.r1.observedLength) return x.r1.observedLength < y.r1.observedLength;
        if (x.r2.observedLength != y.r2.observedLength) return x.r2.observedLength < y.r2.observedLength;
        return a < b;
    }
};
struct PosIdxLess { const i64 *v; ISAAC_HD bool operator()(u16 a, u16 b) const { return v[a] < v[b]; } };

struct BestPairInfo
{
    u8 *frags[2]; u32 cap; u32 n[2]; u32 overflow;
    double bestTemplateLogProbability; u64 bestTemplateScore; u32 resolvedTemplateCount, bestPairEditDistance; double totalTemplateProbability;
    ISAAC_HD void clear()
    { bestTemplateLogProbability = -1.7976931348623157e308; bestTemplateScore = ~u64(0); resolvedTemplateCount = 0; bestPairEditDistance = 0; totalTemplateProbability = 0.0; n[0] = n[1] = 0; overflow = 0; }
    ISAAC_HD void push(u32 r, u32 idx) { if (n[r] < cap) frags[r][n[r]++] = u8(idx); else overflow = 1; }
    ISAAC_HD void init(u32 a, u32 b) { clear(); push(0, a); push(1, b); }
};

// per-thread scratch of the template stage: pointers into one arena (templateWorkBytes / templateWorkBind)
struct TemplateWork
{
    TemplateCaps caps;
    Cand *shadowList; u32 nShadows;
    u32 *shadowCigar;
    i64 *candidatePositions;
    u32 *kmerTable; u32 kmerGeneration;
    ShadowProb *shadowProbs[2]; u32 nShadowProbs[2];
    PairProb *pairProbs; u32 nPairProbs;
    u16 *sortIdx;
    double *terms;            // exp() of the sorted probabilities (fast path of sumUnique*Probabilities)
    Cand *bestOrphanShadows[2]; u32 nBestOrphanShadows[2];
    u32 *templateCigar; u32 templateCigarUsed;
    u32 *tflags;
    BestPairInfo bestCombination, bestRescued;
    u32 overflow;
    u32 lastPushes, lastTruncated;   // of the last findShadowCandidatePositions: list length before sort/unique, capacity hit
};
ISAAC_HD u64 alignUp(u64 v) { return (v + 15) & ~u64(15); }
ISAAC_HD u64 templateWorkBytes(const TemplateCaps &c)
{
    const u64 sortN = imax(imax(c.pos, c.prob), c.pair);
    return alignUp(u64(c.shadow) * sizeof(Cand)) + alignUp(u64(c.shadowCigar) * 4) + alignUp(u64(c.pos) * 8) + alignUp(u64(c.kmerTable) * 4) +
           2 * alignUp(u64(c.prob) * sizeof(ShadowProb)) + alignUp(u64(c.pair) * sizeof(PairProb)) + alignUp(sortN * 2) + alignUp(sortN * 8) + 2 * alignUp(u64(c.best) * sizeof(Cand)) +
           alignUp(u64(c.templateCigar) * 4) + alignUp(u64(c.tflags) * 4) + 4 * alignUp(u64(c.best));
}
// binds the pointers of `w` into the arena at `base` (16-byte aligned, templateWorkBytes(caps) long); the k-mer table must be
// zero at first use (generation 0)
ISAAC_HD void templateWorkBind(TemplateWork &w, void *base, const TemplateCaps &c)
{
    u8 *p = static_cast<u8 *>(base);
    const u64 sortN = imax(imax(c.pos, c.prob), c.pair);
    w.caps = c;
    w.shadowList = reinterpret_cast<Cand *>(p); p += alignUp(u64(c.shadow) * sizeof(Cand));
    w.shadowCigar = reinterpret_cast<u32 *>(p); p += alignUp(u64(c.shadowCigar) * 4);
    w.candidatePositions = reinterpret_cast<i64 *>(p); p += alignUp(u64(c.pos) * 8);
    w.kmerTable = reinterpret_cast<u32 *>(p); p += alignUp(u64(c.kmerTable) * 4);
    for (u32 i = 0; i < 2; ++i) { w.shadowProbs[i] = reinterpret_cast<ShadowProb *>(p); p += alignUp(u64(c.prob) * sizeof(ShadowProb)); }
    w.pairProbs = reinterpret_cast<PairProb *>(p); p += alignUp(u64(c.pair) * sizeof(PairProb));
    w.sortIdx = reinterpret_cast<u16 *>(p); p += alignUp(sortN * 2);
    w.terms = reinterpret_cast<double *>(p); p += alignUp(sortN * 8);
    for (u32 i = 0; i < 2; ++i) { w.bestOrphanShadows[i] = reinterpret_cast<Cand *>(p); p += alignUp(u64(c.best) * sizeof(Cand)); }
    w.templateCigar = reinterpret_cast<u32 *>(p); p += alignUp(u64(c.templateCigar) * 4);
    w.tflags = reinterpret_cast<u32 *>(p); p += alignUp(u64(c.tflags) * 4);
    for (u32 i = 0; i < 2; ++i) { w.bestCombination.frags[i] = p; p += alignUp(u64(c.best)); }
    for (u32 i = 0; i < 2; ++i) { w.bestRescued.frags[i] = p; p += alignUp(u64(c.best)); }
    w.bestCombination.cap = c.best; w.bestRescued.cap = c.best;
    w.kmerGeneration = 0xff; // forces a clear of the table on first use
    w.nShadows = 0; w.nShadowProbs[0] = w.nShadowProbs[1] = 0; w.nPairProbs = 0; w.nBestOrphanShadows[0] = w.nBestOrphanShadows[1] = 0;
    w.templateCigarUsed = 0; w.overflow = 0;
}

// BamTemplate: two fragments, each with a pointer to the buffer its CIGAR lives in
struct Frag { Cand c; const u32 *pool; };
struct BamTemplate { Frag f[2]; u32 n; u32 alignmentScore; bool properPair; };

// What TemplateBuilder's loops over the orphans (TemplateBuilder.cpp:527-590, :742-824) read of a problem's best shadow: whether there is one
// (`rescued`: the problem is valid and ShadowAligner::rescueShadow returned true), its logProbability, Smith-Waterman score and edit distance, and
// what isVeryBadAlignment (TemplateBuilder.cpp:52-62) looks at: the bases its CIGAR maps, its mismatches and its longest run of matches.
struct RescueOutcome { double logProbability; u32 smithWatermanScore; u16 editDistance, mapped, mismatchCount, matchesInARow; u8 rescued; u8 pad[3]; };
static_assert(sizeof(RescueOutcome) == 24, "RescueOutcome layout");

// One mate-rescue problem (ShadowAligner::rescueShadow call): planned by the cluster's thread, its window scanned by a
// wavefront (k_rescue_windows), its candidate positions aligned one per thread (k_rescue_align), consumed by the cluster again.
struct RescueJob
{
    i64 windowBegin;        // first reference base of the scanned window (candidatePositionOffset of ShadowAligner.cpp:190)
    u32 windowLen;          // bases in [windowBegin, windowEnd)
    u32 cluster;            // index inside the chunk
    u32 contigId;
    u32 candBase, nCands;   // slots of the unique candidate start positions (ascending)
    u32 pushes;             // entries shadowCandidatePositions_ would have held before sort/unique (capacity 10000)
    u32 bitmapBase, bitmapWords;
    u8 shadowReadIndex, shadowReverse, valid, fallback;   // fallback: redo this job serially (capacity exceeded)
    u32 gappedBase, nGapped; // the job's gapped retries in the chunk's GappedResult array; gappedBase 0xffffffff: run them serially
    u32 nAligned, bestRank, bestSlot, lastAligned;   // summarizeRescueJob: aligned candidates, the best of them, "the last candidate aligned"
    // finishRescueFlat (k_cluster_sums): what ShadowAligner::rescueShadow leaves behind, without the list itself
    u32 take;               // shadows in the list (min(nAligned, 1000))
    u32 finalBestRank;      // rank of the best shadow after the gapped retries (it is swapped to the front of the list)
    u32 finalBestSlot;      // its candidate slot
    u32 finalBestGapped;    // index of the retry that produced it in the chunk's GappedResult array, 0xffffffff: its ungapped alignment stands
    u8 orphanListIndex;     // the orphan is candidate orphanListIndex of read 1 - shadowReadIndex
    u8 rescued;             // rescueShadow's return value
    u16 windowBaseHigh;     // index of the window's first base in the concatenated contigs (contig offset + windowBegin), 48 bits:
    u32 windowBaseLow;      // k_rescue_windows starts its loads from the job record alone
    RescueOutcome out;      // finishRescueFlat: what the template stage asks about the best shadow, so that it need not follow the record to the shadow
    u32 adapterRange;       // the shadow strand's sequencing adapter, decided by the window's first candidate (ShadowAligner.cpp:207,222); 0: none
    u32 pad;
};
ISAAC_HD u64 rescueJobWindowBase(const RescueJob &j) { return (u64(j.windowBaseHigh) << 32) | j.windowBaseLow; }
static_assert(sizeof(RescueJob) == 128 && offsetof(RescueJob, out) == 96, "RescueJob layout");
static const u32 SHADOW_LIST_MAX = 1000;          // ShadowAligner.hh: shadowList_ capacity, TemplateBuilder.hh:TRACKED_REPEATS_MAX_ONE_READ
enum { RESCUE_SERIAL = 0, RESCUE_PLAN = 1, RESCUE_LOOKUP = 2, RESCUE_PRECOMPUTED = 3 };
// Per cluster, from k_cluster_sums: sumUniqueShadowProbabilities of either side (the shadows rescued by the orphans of read `side`
// plus the seeded candidates of the other read), sumUniquePairProbabilities, and the running sum of TemplateBuilder::rescueShadow
// (TemplateBuilder.cpp:552-560: exp(orphan + shadow) over every shadow of every orphan, in list order).
struct ClusterSums { double shadow[2]; double pair; double ordered; };
static const u32 SHADOW_POSITIONS_MAX = 10000;   // ShadowAligner.hh:91

struct TemplateCtx
{
    const DevParams *P; const DevReference *R; const DevTls *tls;
    u32 rescueMode; u32 jobNext, jobCount; RescueJob *jobs; bool planWrite; bool serialFallbackAllowed;
    const i32 *candPositions; const Cand *shadowCands; const u32 *shadowCigars;   // RESCUE_LOOKUP inputs
    const GappedResult *gappedResults; const GappedJob *gappedJobs;               // RESCUE_LOOKUP: the chunk's gapped retries, or NULL
    const u32 *candRank;                                                          // RESCUE_LOOKUP: aligned candidates before each slot (summarizeRescueJob)
    const ClusterSums *sums;                                                      // RESCUE_PRECOMPUTED
    // the best shadow of the last successful shadowRescue and the buffer its CIGAR lives in (shadowList[0] / shadowCigar in the
    // list-building modes, a private copy / the flat pass's arrays in RESCUE_PRECOMPUTED)
    const Cand *bestRescued; const u32 *bestRescuedPool; Cand bestRescuedCopy;
    ReadView reads[2];
    const ClusterFragments *frags;
    const Cand *cands[2]; u32 nCands[2];      // the cluster's candidate lists: frags->cands, or a private copy of short lists (templateCtxInit)
    TemplateWork *w;
    double rogRead[2], rog;
    u32 clusterId;
    Counters *cnt;
    // wave-cooperative form (k_select_heavy): all `lanes` lanes of the wave run the same statements on the same arena; the bulk
    // loops are strided by `lane`.  lanes == 1: plain thread-serial execution.  fastSort: large probability lists are sorted by a
    // total order that refines the reference's comparators, with the exact std::sort replica as fallback for ambiguous data.
    u32 lanes, lane; bool fastSort; u16 *ldsSort; u32 ldsSortCap;
    u32 mapqNearInteger;      // mapqFloor met an argument within 1e-11 of an integer for this cluster (RECORD_MAPQ_NEAR_INTEGER)
    long long prof[8];
};

// optional section timers of the heavy path (-DISAAC_PROFILE_HEAVY): shader clock ticks per section, printed per cluster
#if defined(ISAAC_PROFILE_HEAVY) && defined(__HIP_DEVICE_COMPILE__)
#define ISAAC_PROF_T0(x) const long long prof_t0 = clock64()
#define ISAAC_PROF_ADD(x, slot) (x).prof[slot] += clock64() - prof_t0
#else
#define ISAAC_PROF_T0(x)
#define ISAAC_PROF_ADD(x, slot)
#endif

// wave-level helpers of the cooperative form; identities in the thread-serial form
ISAAC_HD bool coopAny(const TemplateCtx &x, bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (x.lanes > 1) return __ballot(v) != 0;
#endif
    (void)x; return v;
}
ISAAC_HD void coopSync(const TemplateCtx &x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (x.lanes > 1) __syncthreads();   // the workgroup is this one wave: orders the lanes' global and LDS traffic
#endif
    (void)x;
}

// Sorts idx[0..n) by a strict TOTAL order.  Cooperative form: bitonic network over the indices in LDS, keys stay where they are.
template <typename LessT>
ISAAC_HD void sortTotal(TemplateCtx &x, u16 *idx, u32 n, LessT less)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (x.lanes > 1)
    {
        u32 m = 1; while (m < n) m <<= 1;
        if (m <= x.ldsSortCap)
        {
            u16 *l = x.ldsSort;
            for (u32 i = x.lane; i < m; i += x.lanes) l[i] = i < n ? idx[i] : u16(0xffff);
            __syncthreads();
            for (u32 k = 2; k <= m; k <<= 1)
                for (u32 j = k >> 1; j > 0; j >>= 1)
                {
                    for (u32 t = x.lane; t < (m >> 1); t += x.lanes)
                    {
                        const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const u16 a = l[i], b = l[i + j];
                        // 0xffff pads the network: greater than every element
                        const bool aLessB = (a != 0xffff) && (b == 0xffff || less(a, b));
                        const bool bLessA = (b != 0xffff) && (a == 0xffff || less(b, a));
                        const bool swap = (0 == (i & k)) ? bLessA : aLessB;
                        if (swap) { l[i] = b; l[i + j] = a; }
                    }
                    __syncthreads();
                }
            for (u32 i = x.lane; i < n; i += x.lanes) idx[i] = l[i];
            __syncthreads();
            return;
        }
    }
#endif
    (void)x;
    exactSort(idx, i32(n), less);   // any correct algorithm will do for a total order
}

ISAAC_HD u32 mapqFloor(TemplateCtx &x, double ratio)
{
    // unsigned(floor(-10.0 * log10(ratio))) of TemplateBuilder.cpp:273,437,604,608,912,916,920.  ratio is in (0, 1], so v >= 0.
    const double v = -10.0 * log10(ratio);
    const double fl = floor(v);
    // exp/log10 of the device maths library and of glibc may differ in the last ulp, i.e. by ~1e-15 in v: the floor can differ
    // only if v sits within that distance of an integer.  Such cases are counted (generously, 1e-11) so that a run can
    // prove it had none; v in [0, 1e-11) is safe because v cannot be negative.
    const double d = v - fl;
    if (d > 1.0 - 1e-11 || (d < 1e-11 && fl >= 1.0))
    {
        ++x.cnt->mapqNearInteger; x.mapqNearInteger = 1;
#ifdef ISAAC_DEBUG_MAPQ
        printf("mapq near integer: ratio=%.17g v=%.17g floor=%.17g cluster=%u\n", ratio, v, fl, x.clusterId);
#endif
    }
    return u32(fl);
}

// ------------------------------------------------------------------------------------------------------------------
// ShadowAligner::findShadowCandidatePositions (ShadowAligner.cpp:53-112).  The reference keeps a 16384-entry direct table of
// the first read position of every 7-mer; the same map is held here in a small generation-tagged hash table.
ISAAC_HD u32 baseCode(char c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }

ISAAC_HD u32 findShadowCandidatePositions(TemplateCtx &x, const char *reference, i64 windowBegin, i64 windowEnd, const ReadView &shadow, bool shadowReverse)
{
    TemplateWork &w = *x.w;
    if (++w.kmerGeneration > 0xff) { for (u32 i = 0; i < KMER_TABLE; ++i) w.kmerTable[i] = 0; w.kmerGeneration = 1; }
    const u32 gen = w.kmerGeneration << 24;
    {
        u32 kmer = 0, valid = 0;
        for (u32 i = 0; i < shadow.length; ++i)
        {
            const char c = strandBase(shadow, shadowReverse, i);
            const u32 v = c == 'n' ? 4u : baseCode(c);
            if (v > 3) { valid = 0; kmer = 0; continue; }
            kmer = ((kmer << 2) | v) & 0x3fff; ++valid;
            if (valid >= 7)
            {
                const u32 pos = i + 1 - 7;
                u32 h = (kmer * 2654435761u) >> 22;
                while (true)
                {
                    const u32 e = w.kmerTable[h];
                    if ((e & 0xff000000u) != gen) { w.kmerTable[h] = gen | (kmer << 10) | pos; break; }
                    if (((e >> 10) & 0x3fff) == kmer) break;      // keep the first occurrence
                    h = (h + 1) & (KMER_TABLE - 1);
                }
            }
        }
    }
    u32 n = 0; bool truncated = false;
    u32 kmer = 0, valid = 0;
    for (i64 p = windowBegin; p < windowEnd; ++p)
    {
        const u32 v = baseCode(reference[p]);
        if (v > 3) { valid = 0; kmer = 0; continue; }
        kmer = ((kmer << 2) | v) & 0x3fff; ++valid;
        if (valid < 7) continue;
        u32 h = (kmer * 2654435761u) >> 22;
        while (true)
        {
            const u32 e = w.kmerTable[h];
            if ((e & 0xff000000u) != gen) break;
            if (((e >> 10) & 0x3fff) == kmer)
            {
                const i64 candidatePosition = (p + 1 - 7 - windowBegin) - i64(e & 0x3ff);
                if (!n || w.candidatePositions[n - 1] != candidatePosition)
                {
                    if (n == w.caps.pos) { truncated = true; break; }
                    w.candidatePositions[n++] = candidatePosition;
                }
                break;
            }
            h = (h + 1) & (KMER_TABLE - 1);
        }
        if (truncated) break;
    }
    if (truncated && w.caps.pos < SHADOW_POSITIONS_MAX) w.overflow = 1;  // the reference itself stops at 10000 positions
    w.lastPushes = n; w.lastTruncated = truncated ? 1 : 0;
    if (n)
    {
        for (u32 i = 0; i < n; ++i) w.sortIdx[i] = u16(i);
        PosIdxLess less; less.v = w.candidatePositions;
        exactSort(w.sortIdx, i32(n), less);
        // std::unique: keep the first index of every run of equal values; sortIdx[0..n) then lists the unique positions ascending
        i64 prev = 0; u32 m = 0;
        for (u32 i = 0; i < n; ++i)
        {
            const i64 val = w.candidatePositions[w.sortIdx[i]];
            if (!i || val != prev) { w.sortIdx[m++] = w.sortIdx[i]; prev = val; }
        }
        n = m;
    }
    return n;
}

// ShadowAligner.cpp:119-149
ISAAC_HD void calculateShadowRescueRange(const TemplateCtx &x, const Cand &orphan, i64 bestTemplateLength, i64 &first, i64 &second)
{
    const u32 shadowReadIndex = (orphan.readIndex + 1) % 2;
    const u32 readLengths[2] = { x.reads[0].length, x.reads[1].length };
    i64 shadowMinPosition = tlsMateMinPosition(*x.tls, orphan.readIndex, orphan.reverse, orphan.position, readLengths);
    i64 shadowMaxPosition = tlsMateMaxPosition(*x.tls, orphan.readIndex, orphan.reverse, orphan.position, readLengths) + i64(readLengths[shadowReadIndex]) - 1;
    if (bestTemplateLength)
    {
        const i64 fpos = i64(refposPosition(candFStrandPos(orphan))), rpos = i64(refposPosition(candRStrandPos(orphan)));
        if (shadowMinPosition < fpos) shadowMinPosition = imin(rpos - bestTemplateLength, shadowMinPosition);
        if (shadowMaxPosition > fpos) shadowMaxPosition = imax(fpos + bestTemplateLength, shadowMaxPosition);
    }
    first = shadowMinPosition - 10; second = shadowMaxPosition + 10;
}

// first half of ShadowAligner::rescueShadow (ShadowAligner.cpp:155-197): where the mate is expected
ISAAC_HD bool planRescue(const TemplateCtx &x, const Cand &orphan, i64 bestTemplateLength, RescueJob &job)
{
    job.windowBegin = 0; job.windowLen = 0; job.cluster = x.clusterId; job.contigId = orphan.contigId; job.candBase = 0; job.nCands = 0; job.pushes = 0;
    job.bitmapBase = 0; job.bitmapWords = 0; job.valid = 0; job.fallback = 0; job.gappedBase = 0xffffffffu; job.nGapped = 0; job.nAligned = 0; job.bestRank = 0; job.bestSlot = 0; job.lastAligned = 0;
    job.adapterRange = 0; job.pad = 0; job.take = 0; job.finalBestRank = 0; job.finalBestSlot = 0; job.finalBestGapped = 0xffffffffu; job.rescued = 0; job.windowBaseHigh = 0; job.windowBaseLow = 0;
    job.orphanListIndex = u8(&orphan - x.cands[orphan.readIndex]);
    job.shadowReadIndex = u8((orphan.readIndex + 1) % 2);
    job.shadowReverse = 0;
    if (!tlsIsCoherent(*x.tls)) return false;
    job.shadowReverse = u8(tlsMateOrientation(*x.tls, orphan.readIndex, orphan.reverse));
    i64 rangeFirst, rangeSecond;
    calculateShadowRescueRange(x, orphan, bestTemplateLength, rangeFirst, rangeSecond);
    if (rangeSecond < rangeFirst) return false;
    if (rangeSecond + 1 + i64(x.reads[job.shadowReadIndex].length) < 0) return false;
    const i64 referenceSize = i64(contigLength(*x.R, orphan.contigId));
    job.windowBegin = imax<i64>(0, rangeFirst);
    const i64 windowEnd = imin(referenceSize, rangeSecond + 1);
    job.windowLen = windowEnd > job.windowBegin ? u32(windowEnd - job.windowBegin) : 0;
    const u64 windowBase = x.R->contigOffset[orphan.contigId] + u64(job.windowBegin);
    job.windowBaseHigh = u16(windowBase >> 32); job.windowBaseLow = u32(windowBase);
    job.valid = 1;
    return true;
}

// second half (ShadowAligner.cpp:232-291): gapped retries next to close candidates, best shadow to the front
// `gapped`: results of the retries in list order when the flat pass already ran them (planRescueGapped), else NULL
ISAAC_HD bool finishRescue(TemplateCtx &x, CigarPool &pool, i32 best, const GappedResult *gapped = 0, u32 adapterRange = 0)
{
    TemplateWork &w = *x.w;
    const DevParams &P = *x.P; const DevReference &R = *x.R;
    if (best < 0) { if (pool.overflow) w.overflow = 1; return false; }
    ISAAC_PROF_T0(x);
    const ReadView &shadowRead = x.reads[w.shadowList[0].readIndex];
    if (BSW_MISMATCHES_CUTOFF < w.shadowList[best].mismatchCount)
    {
        for (u32 i = 0; i < w.nShadows; ++i)
        {
            Cand &fragment = w.shadowList[i];
            if (i + 1 != w.nShadows && w.shadowList[i + 1].position - fragment.position < i64(BSW_DISTANCE_CUTOFF))
            {
                if (BSW_MISMATCHES_CUTOFF < fragment.mismatchCount)
                {
                    Cand tmp = fragment;
                    ++x.cnt->rescueBsw;
                    u32 matchCount;
                    if (gapped)
                    {
                        const GappedResult &g = *gapped++;
                        tmp = g.out; matchCount = g.matchCount; tmp.cigarOffset = pool.used;
                        if (0xffffffffu == g.nCigar) { w.overflow = 1; matchCount = 0; }
                        else for (u32 k = 0; k < g.nCigar; ++k) pool.push(g.cigar[k]);
                    }
                    else matchCount = alignGapped(P, R, shadowRead, tmp, pool, w.tflags, adapterRange);
                    if (matchCount && matchCount + BSW_WIDEST_GAP_SIZE > candObservedLength(fragment) && (tmp.mismatchCount <= P.gappedMismatchesMax) &&
                        (fragment.mismatchCount > tmp.mismatchCount) && lpLess(fragment.logProbability, tmp.logProbability))
                    {
                        fragment = tmp;
                        if (lpLess(w.shadowList[best].logProbability, fragment.logProbability)) best = i32(i);
                    }
                }
            }
        }
    }
    if (pool.overflow) w.overflow = 1;
    if (best != 0) { const Cand t = w.shadowList[0]; w.shadowList[0] = w.shadowList[best]; w.shadowList[best] = t; }
    ISAAC_PROF_ADD(x, 1);
    return true;
}

// finishRescue when the retries were planned and run by the flat pass: only the planned elements are visited (in list
// order, like the loop above), found through the rank of their candidate slot
ISAAC_HD bool finishRescueLookup(TemplateCtx &x, CigarPool &pool, i32 best, const RescueJob &job)
{
    TemplateWork &w = *x.w;
    const DevParams &P = *x.P;
    if (best < 0) { if (pool.overflow) w.overflow = 1; return false; }
    ISAAC_PROF_T0(x);
    for (u32 k = 0; k < job.nGapped; ++k)
    {
        const GappedResult &g = x.gappedResults[job.gappedBase + k];
        const u32 i = x.candRank[x.gappedJobs[job.gappedBase + k].tag];
        Cand &fragment = w.shadowList[i];
        ++x.cnt->rescueBsw;
        Cand tmp = g.out; u32 matchCount = g.matchCount; tmp.cigarOffset = pool.used;
        if (0xffffffffu == g.nCigar) { w.overflow = 1; matchCount = 0; }
        else for (u32 c = 0; c < g.nCigar; ++c) pool.push(g.cigar[c]);
        if (matchCount && matchCount + BSW_WIDEST_GAP_SIZE > candObservedLength(fragment) && (tmp.mismatchCount <= P.gappedMismatchesMax) &&
            (fragment.mismatchCount > tmp.mismatchCount) && lpLess(fragment.logProbability, tmp.logProbability))
        {
            fragment = tmp;
            if (lpLess(w.shadowList[best].logProbability, fragment.logProbability)) best = i32(i);
        }
    }
    if (pool.overflow) w.overflow = 1;
    if (best != 0) { const Cand t = w.shadowList[0]; w.shadowList[0] = w.shadowList[best]; w.shadowList[best] = t; }
    ISAAC_PROF_ADD(x, 1);
    return true;
}

// ShadowAligner::rescueShadow, everything in this thread (RESCUE_SERIAL)
ISAAC_HD bool shadowRescueSerial(TemplateCtx &x, const Cand &orphan, const RescueJob &job)
{
    TemplateWork &w = *x.w;
    const DevParams &P = *x.P; const DevReference &R = *x.R;
    const ReadView &shadowRead = x.reads[job.shadowReadIndex];
    const char *reference = R.bases + R.contigOffset[orphan.contigId];
    const u32 nPositions = findShadowCandidatePositions(x, reference, job.windowBegin, job.windowBegin + job.windowLen, shadowRead, job.shadowReverse != 0);
    x.cnt->rescueCandidates += nPositions;
    CigarPool pool; pool.words = w.shadowCigar; pool.used = 0; pool.capacity = w.caps.shadowCigar; pool.overflow = 0;
    i32 best = -1;
    u32 adapterRange = 0;                // a fresh FragmentSequencingAdapterClipper per rescue: the first candidate position decides (ShadowAligner.cpp:207,222)
    for (u32 c = 0; c < nPositions; ++c)
    {
        if (w.nShadows == w.caps.shadow) { if (w.caps.shadow < SHADOW_LIST_MAX) w.overflow = 1; return false; } // reference: capacity 1000 -> return false
        Cand &fragment = w.shadowList[w.nShadows];
        candInit(fragment, job.shadowReadIndex);
        fragment.reverse = job.shadowReverse; fragment.contigId = orphan.contigId;
        fragment.position = w.candidatePositions[w.sortIdx[c]] + job.windowBegin;
        if (0 == c && P.adapters)
        {
            ReadView whole = shadowRead; whole.endCyclesMasked = 0;
            adapterRange = adapterStrandRange(*P.adapters, R, whole, 0 != job.shadowReverse, orphan.contigId, fragment.position);
        }
        ++x.cnt->ungappedScans;
        if (alignUngapped(P, R, shadowRead, fragment, pool, adapterRange))
        {
            if (best < 0 || lpLess(w.shadowList[best].logProbability, fragment.logProbability)) best = i32(w.nShadows);
            ++w.nShadows;
        }
    }
    return finishRescue(x, pool, best, 0, adapterRange);
}

// ShadowAligner::rescueShadow with the window scan and the ungapped alignments already done by the flat kernels (RESCUE_LOOKUP)
ISAAC_HD bool shadowRescueLookup(TemplateCtx &x, const Cand &orphan, const RescueJob &job)
{
    TemplateWork &w = *x.w;
    if (job.fallback || (job.nGapped && (0xffffffffu == job.gappedBase || !x.gappedResults || !x.gappedJobs)))
    {   // a capacity of the flat pass was exceeded for this job: exact serial path, which needs the reference-sized lists
        if (!x.serialFallbackAllowed) { w.overflow = 1; return false; }
        return shadowRescueSerial(x, orphan, job);
    }
    CigarPool pool; pool.words = w.shadowCigar; pool.used = 0; pool.capacity = w.caps.shadowCigar; pool.overflow = 0;
    ISAAC_PROF_T0(x);
    // The reference appends the aligned candidates one by one and gives up ("return false", list kept) when the list is full
    // and another candidate turns up.  With the ranks known (summarizeRescueJob) the copies are independent of each other.
    const u32 K = w.caps.shadow;
    const bool full = job.nAligned - job.lastAligned >= K && job.nCands != 0;
    const u32 take = imin(job.nAligned, K);
    for (u32 c = x.lane; c < job.nCands; c += x.lanes)
    {
        const u32 slot = job.candBase + c;
        const u32 rank = x.candRank[slot];
        const Cand &src = x.shadowCands[slot];
        if (rank >= take || !candAligned(src)) continue;
        Cand &fragment = w.shadowList[rank];
        fragment = src;
        fragment.cigarOffset = 0; fragment.cigarLength = src.cigarLength;   // only the best shadow's CIGAR is ever read: it is put in place below
    }
    coopSync(x);
    w.nShadows = take;
    if (full) { if (K < SHADOW_LIST_MAX) w.overflow = 1; ISAAC_PROF_ADD(x, 0); return false; }
    i32 best = -1;
    if (job.nAligned)
    {
        best = i32(job.bestRank);
        Cand &b = w.shadowList[best];
        b.cigarOffset = pool.used;
        for (u32 k = 0; k < b.cigarLength; ++k) pool.push(x.shadowCigars[u64(job.bestSlot) * 3 + k]);
    }
    ISAAC_PROF_ADD(x, 0);
    return finishRescueLookup(x, pool, best, job);
}

// What the walks below want of a rescue candidate, 16 bytes instead of the 64-byte record (round 6: k_rescue_align writes one beside every candidate; a thread
// that walks a problem's list read a 64-byte line per candidate for these fields -- 1.1 GB a launch of k_rescue_gapped_plan).  The position is relative to the
// problem's window (only differences between the candidates of one problem are looked at).
struct CandSummary { double logProbability; i32 relativePosition; u16 mismatchCount, cigarLength; };
static_assert(sizeof(CandSummary) == 16, "CandSummary layout");
ISAAC_HD CandSummary candSummary(const Cand &c, i64 windowBegin)
{ CandSummary s; s.logProbability = c.logProbability; s.relativePosition = i32(c.position - windowBegin); s.mismatchCount = c.mismatchCount; s.cigarLength = c.cigarLength; return s; }
// the fields of candidate `slot` whichever way they are kept (summaries == NULL: read from the records, rounds 1-5 and the host forms)
ISAAC_HD void candSummaryFields(const Cand *shadowCands, const CandSummary *summaries, u32 slot, double &lp, i64 &position, u32 &mismatches, bool &aligned)
{
    if (summaries) { const CandSummary s = summaries[slot]; lp = s.logProbability; position = s.relativePosition; mismatches = s.mismatchCount; aligned = 0 != s.cigarLength; }
    else { const Cand &f = shadowCands[slot]; lp = f.logProbability; position = f.position; mismatches = f.mismatchCount; aligned = candAligned(f); }
}
// One pass over a job's aligned candidates (in candidate order): how many there are before each slot, which is the best
// (the running choice of ShadowAligner.cpp:216-230) and whether the last candidate aligned.
static const u32 SUMMARY_BATCH = 8;
ISAAC_HD void summarizeRescueJob(RescueJob &job, const Cand *shadowCands, u32 *candRank, const CandSummary *summaries = 0)
{
    u32 n = 0; i32 best = -1; u32 bestRank = 0, bestMismatches = 0; bool last = false;
    double bestLp = 0.0;
    u32 nClose = 0; i64 prevPosition = 0; u32 prevMismatches = 0;     // the pairs planRescueGapped would pick, counted on the way
    // (the record's two numbers once: read through `job` inside the loops they were fetched again after every store to candRank, which may alias it for all
    // the compiler knows -- two more dependent loads in front of every batch of a walk that is nothing but dependent loads)
    const u32 nCands = job.nCands, candBase = job.candBase;
    // the candidates' fields are fetched SUMMARY_BATCH at a time before any of them is looked at: one thread walks the whole
    // list (thousands of entries in repeat families), and loads that wait for the previous element's branches cost a memory
    // latency each
    for (u32 c0 = 0; c0 < nCands; c0 += SUMMARY_BATCH)
    {
        double lps[SUMMARY_BATCH]; i64 positions[SUMMARY_BATCH]; u32 mismatchCounts[SUMMARY_BATCH]; bool aligned[SUMMARY_BATCH];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (u32 k = 0; k < SUMMARY_BATCH; ++k) candSummaryFields(shadowCands, summaries, candBase + imin(c0 + k, nCands - 1), lps[k], positions[k], mismatchCounts[k], aligned[k]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (u32 k = 0; k < SUMMARY_BATCH; ++k)
        {
            const u32 c = c0 + k;
            if (c >= nCands) break;
            candRank[candBase + c] = n;
            last = aligned[k];
            if (!last) continue;
            const double lp = lps[k]; const i64 position = positions[k]; const u32 mismatches = mismatchCounts[k];
            if (best < 0 || lpLess(bestLp, lp)) { best = i32(c); bestRank = n; bestLp = lp; bestMismatches = mismatches; }
            if (n && position - prevPosition < i64(BSW_DISTANCE_CUTOFF) && BSW_MISMATCHES_CUTOFF < prevMismatches) ++nClose;
            prevPosition = position; prevMismatches = mismatches;
            ++n;
        }
    }
    job.nAligned = n; job.bestRank = bestRank; job.bestSlot = best < 0 ? 0 : candBase + u32(best); job.lastAligned = last ? 1 : 0;
    job.nGapped = (best >= 0 && BSW_MISMATCHES_CUTOFF < bestMismatches) ? nClose : 0;   // == planRescueGapped(job, ..., NULL)
}

// Which shadows of a job ShadowAligner.cpp:232-262 hands to the gapped aligner.  The choice reads only the ungapped results
// (the loop compares each element with its not yet modified successor), so the retries can run before the cluster's thread
// consumes them.  out == NULL: count only.
// The second walk of planRescueGapped for a problem whose summary says that there are retries (job.nGapped, so the best candidate's mismatches are
// known to qualify): positions, mismatch counts and "aligned" fetched SUMMARY_BATCH candidates at a time, as in the summary -- the few threads of a wave
// that come here would otherwise walk their lists a load at a time, twice, while the others wait.
ISAAC_HD u32 writeRescueGapped(const RescueJob &job, const Cand *shadowCands, const u32 *shadowCigars, u32 endCyclesMasked, GappedJob *out, const CandSummary *summaries = 0)
{
    u32 n = 0; i32 prev = -1; i64 prevPosition = 0; u32 prevMismatches = 0;
    const u32 nCands = job.nCands, candBase = job.candBase;            // (once: the stores to `out` may alias the record)
    for (u32 c0 = 0; c0 < nCands; c0 += SUMMARY_BATCH)
    {
        i64 positions[SUMMARY_BATCH]; u32 mismatchCounts[SUMMARY_BATCH]; bool aligned[SUMMARY_BATCH];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (u32 k = 0; k < SUMMARY_BATCH; ++k) { double lp; candSummaryFields(shadowCands, summaries, candBase + imin(c0 + k, nCands - 1), lp, positions[k], mismatchCounts[k], aligned[k]); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (u32 k = 0; k < SUMMARY_BATCH; ++k)
        {
            const u32 c = c0 + k;
            if (c >= nCands) break;
            if (!aligned[k]) continue;
            if (prev >= 0 && positions[k] - prevPosition < i64(BSW_DISTANCE_CUTOFF) && BSW_MISMATCHES_CUTOFF < prevMismatches)
            {
                GappedJob &g = out[n]; g.in = shadowCands[candBase + u32(prev)]; g.cluster = job.cluster; g.endCyclesMasked = u16(endCyclesMasked); g.accepted = 0; g.tag = candBase + u32(prev); g.adapterRange = job.adapterRange;
                g.in.cigarOffset = 0;
                g.in.position = candUnclippedPosition(g.in, shadowCigars + u64(candBase + u32(prev)) * 3); g.in.cigarLength = 0;
                ++n;
            }
            prev = i32(c); prevPosition = positions[k]; prevMismatches = mismatchCounts[k];
        }
    }
    return n;
}
ISAAC_HD u32 planRescueGapped(const RescueJob &job, const Cand *shadowCands, const u32 *shadowCigars, u32 endCyclesMasked, GappedJob *out)
{
    i32 best = -1;
    for (u32 c = 0; c < job.nCands; ++c)
    {
        const Cand &f = shadowCands[job.candBase + c];
        if (!candAligned(f)) continue;
        if (best < 0 || lpLess(shadowCands[job.candBase + best].logProbability, f.logProbability)) best = i32(c);
    }
    if (best < 0 || !(BSW_MISMATCHES_CUTOFF < shadowCands[job.candBase + best].mismatchCount)) return 0;
    u32 n = 0; i32 prev = -1;
    for (u32 c = 0; c < job.nCands; ++c)
    {
        const Cand &next = shadowCands[job.candBase + c];
        if (!candAligned(next)) continue;
        if (prev >= 0)
        {
            const Cand &f = shadowCands[job.candBase + prev];
            if (next.position - f.position < i64(BSW_DISTANCE_CUTOFF) && BSW_MISMATCHES_CUTOFF < f.mismatchCount)
            {
                if (out)
                {
                    GappedJob &g = out[n]; g.in = f; g.cluster = job.cluster; g.endCyclesMasked = u16(endCyclesMasked); g.accepted = 0; g.tag = job.candBase + u32(prev); g.adapterRange = job.adapterRange;
                    g.in.cigarOffset = 0;
                    g.in.position = candUnclippedPosition(g.in, shadowCigars + u64(job.candBase + u32(prev)) * 3); g.in.cigarLength = 0;
                }
                ++n;
            }
        }
        prev = i32(c);
    }
    return n;
}

// ShadowAligner::rescueShadow (ShadowAligner.cpp:155-291).  Fills w.shadowList (best first when true is returned); in
// RESCUE_PRECOMPUTED there is no list: the flat pass left the outcome in the job.  x.bestRescued / x.bestRescuedPool: the best shadow.
ISAAC_HD bool shadowRescue(TemplateCtx &x, const Cand &orphan, i64 bestTemplateLength)
{
    TemplateWork &w = *x.w;
    w.nShadows = 0;
    x.bestRescued = w.shadowList; x.bestRescuedPool = w.shadowCigar;
    if (RESCUE_PLAN == x.rescueMode)
    {   // only the sequence of rescue problems is wanted: it does not depend on their results
        RescueJob job;
        planRescue(x, orphan, bestTemplateLength, job);
        if (x.planWrite) x.jobs[x.jobNext] = job;
        ++x.jobNext;
        return false;
    }
    if (RESCUE_PRECOMPUTED == x.rescueMode)
    {
        const RescueJob &job = x.jobs[x.jobNext++];
        if (!job.valid || !job.rescued) return false;
        if (0xffffffffu != job.finalBestGapped)
        {
            const GappedResult &g = x.gappedResults[job.finalBestGapped];
            x.bestRescuedCopy = g.out; x.bestRescuedCopy.cigarLength = u16(g.nCigar); x.bestRescuedPool = g.cigar;
        }
        else { x.bestRescuedCopy = x.shadowCands[job.finalBestSlot]; x.bestRescuedPool = x.shadowCigars + u64(job.finalBestSlot) * 3; }
        x.bestRescuedCopy.cigarOffset = 0;
        x.bestRescued = &x.bestRescuedCopy;
        return true;
    }
    if (RESCUE_LOOKUP == x.rescueMode)
    {
        const RescueJob &job = x.jobs[x.jobNext++];
        if (!job.valid) return false;
        return shadowRescueLookup(x, orphan, job);
    }
    RescueJob job;
    if (!planRescue(x, orphan, bestTemplateLength, job)) return false;
    ++x.cnt->rescueCalls; x.cnt->rescueWindowBases += job.windowLen;
    const bool ret = shadowRescueSerial(x, orphan, job);
    return ret;
}

// ------------------------------------------------------------------------------------------------------------------
// TemplateBuilder
ISAAC_HD bool isVeryBadAlignment(const Cand &f, const u32 *pool, double logMismatchQ40)
{
    const u32 mapped = candMappedLength(f, pool);
    return f.matchesInARow < 32 && (u32(f.mismatchCount) > mapped / 8 || f.logProbability < logMismatchQ40 / 4 * mapped);
}

// getBestFragment (TemplateBuilder.cpp:177-226): lowest smithWatermanScore, then highest logProbability (epsilon 1e-7); the
// reference keeps every candidate that ties with the one that last improved the best and returns the first of them, or the
// (clusterId % count)-th with --scatter-repeats
ISAAC_HD u32 getBestFragment(const TemplateCtx &x, u32 r)
{
    const Cand *list = x.cands[r]; const u32 n = x.nCands[r];
    u32 bestScore = 0xffffffffu; double bestLp = -1.7976931348623157e308;
    u32 first = 0, count = 0;
    for (u32 i = 0; i < n; ++i)
    {
        if (bestScore > list[i].smithWatermanScore || (bestScore == list[i].smithWatermanScore && lpLess(bestLp, list[i].logProbability)))
        { bestScore = list[i].smithWatermanScore; bestLp = list[i].logProbability; first = i; count = 1; }
        else if (bestScore == list[i].smithWatermanScore && lpEquals(bestLp, list[i].logProbability)) ++count;
    }
    if (!x.P->scatterRepeats || count < 2) return first;
    u32 want = x.clusterId % count;
    if (!want) return first;
    for (u32 i = first + 1; i < n; ++i)
        if (bestScore == list[i].smithWatermanScore && lpEquals(bestLp, list[i].logProbability) && 0 == --want) return i;
    return first;
}

// updateMappingScore (TemplateBuilder.cpp:233-285)
ISAAC_HD bool updateMappingScore(TemplateCtx &x, Cand &fragment, u32 r, u32 listIndex, bool forceWellAnchored)
{
    if (forceWellAnchored || candWellAnchored(fragment))
    {
        const Cand *list = x.cands[r]; const u32 n = x.nCands[r];
        double neighborProbability = x.rogRead[list[listIndex].readIndex];
        for (u32 i = 0; i < n; ++i) if (listIndex != i) neighborProbability += exp(list[i].logProbability);
        fragment.alignmentScore = mapqFloor(x, neighborProbability / (neighborProbability + exp(list[listIndex].logProbability)));
        return true;
    }
    fragment.alignmentScore = 0;
    return false;
}

// locateBestPair (TemplateBuilder.cpp:287-391)
ISAAC_HD void locateBestPair(TemplateCtx &x, BestPairInfo &ret)
{
    const Cand *l0 = x.cands[0], *l1 = x.cands[1];
    const u32 n0 = x.nCands[0], n1 = x.nCands[1];
    ret.init(0, 0);
    u32 b0 = 0, b1 = 0;
    while (b0 != n0 && b1 != n1)
    {
        u32 e0 = b0 + 1; while (e0 != n0 && l0[e0].contigId == l0[b0].contigId) ++e0;
        u32 e1 = b1 + 1; while (e1 != n1 && l1[e1].contigId == l1[b1].contigId) ++e1;
        if (l0[b0].contigId == l1[b1].contigId)
        {
            for (u32 i = b0; i != e0; ++i) for (u32 j = b1; j != e1; ++j)
            {
                if (tlsMatchModel(*x.tls, l0[i], l1[j]))
                {
                    const double currentLogProbability = l0[i].logProbability + l1[j].logProbability;
                    const double currentProbability = exp(currentLogProbability);
                    const u64 templateScore = u64(l0[i].smithWatermanScore + l1[j].smithWatermanScore);   // unsigned + unsigned, then widened
                    ret.totalTemplateProbability += currentProbability;
                    if (0 == ret.resolvedTemplateCount || ret.bestTemplateScore > templateScore ||
                        (templateScore == ret.bestTemplateScore && lpLess(ret.bestTemplateLogProbability, currentLogProbability)))
                    {
                        ret.n[0] = ret.n[1] = 0; ret.push(0, i); ret.push(1, j);
                        ret.bestTemplateScore = templateScore; ret.bestTemplateLogProbability = currentLogProbability;
                    }
                    else if (templateScore == ret.bestTemplateScore && lpEquals(currentLogProbability, ret.bestTemplateLogProbability)) { ret.push(0, i); ret.push(1, j); }
                    ++ret.resolvedTemplateCount;
                }
            }
            b0 = e0; b1 = e1;
        }
        else if (l0[b0].contigId < l1[b1].contigId) b0 = e0; else b1 = e1;
    }
    if (ret.resolvedTemplateCount) ret.bestPairEditDistance = u32(l0[ret.frags[0][0]].editDistance) + u32(l1[ret.frags[1][0]].editDistance);
}

ISAAC_HD void fragFromList(const TemplateCtx &x, Frag &f, u32 r, u32 idx) { f.c = x.cands[r][idx]; f.pool = x.frags->cigarPool; }

// buildPairedEndTemplate (TemplateBuilder.cpp:398-465)
ISAAC_HD bool buildPairedEndTemplate(TemplateCtx &x, BamTemplate &t, BestPairInfo &best)
{
    if (x.P->scatterRepeats)
    {
        const u32 repeatIndex = x.clusterId % best.n[0];
        u8 s = best.frags[0][0]; best.frags[0][0] = best.frags[0][repeatIndex]; best.frags[0][repeatIndex] = s;
        s = best.frags[1][0]; best.frags[1][0] = best.frags[1][repeatIndex]; best.frags[1][repeatIndex] = s;
    }
    fragFromList(x, t.f[0], 0, best.frags[0][0]);
    fragFromList(x, t.f[1], 1, best.frags[1][0]);
    Cand &read1 = t.f[0].c, &read2 = t.f[1].c;
    const bool r1WellAnchored = updateMappingScore(x, read1, 0, best.frags[0][0], candWellAnchored(read2));
    const bool r2WellAnchored = updateMappingScore(x, read2, 1, best.frags[1][0], candWellAnchored(read1));
    t.properPair = TLS_NOMINAL == tlsCheckModel(*x.tls, read1, read2);
    if (r1WellAnchored || r2WellAnchored)
    {
        const double otherPairsProbability = (best.totalTemplateProbability - exp(best.bestTemplateLogProbability)) + x.rog;
        t.alignmentScore = mapqFloor(x, otherPairsProbability / (best.totalTemplateProbability + x.rog));
        return r1WellAnchored && r2WellAnchored && !read1.repeatSeedsCount && !read2.repeatSeedsCount;
    }
    t.alignmentScore = 0xffffffffu;
    return false;
}

// flagDodgyTemplate (TemplateBuilder.cpp:467-493, 1010-1033)
ISAAC_HD bool flagDodgyTemplate2(const TemplateCtx &x, Cand &orphan, Cand &shadow, BamTemplate &t)
{
    if (-1 == x.P->dodgyAlignmentScore) { candSetNoMatch(orphan); candSetNoMatch(shadow); t.alignmentScore = 0xffffffffu; return false; }
    orphan.alignmentScore = 0xffffffffu; shadow.alignmentScore = 0xffffffffu; t.alignmentScore = 0xffffffffu;
    return true;
}

// cloneWithCigar (TemplateBuilder.cpp:678-687): the clone's CIGAR lives in the template buffer
ISAAC_HD Cand cloneWithCigar(TemplateWork &w, const Cand &right, const u32 *rightPool)
{
    Cand ret = right;
    ret.cigarOffset = w.templateCigarUsed;
    for (u32 i = 0; i < right.cigarLength; ++i)
    {
        if (w.templateCigarUsed < w.caps.templateCigar) w.templateCigar[w.templateCigarUsed++] = rightPool[right.cigarOffset + i];
        else w.overflow = 1;
    }
    return ret;
}

ISAAC_HD void pushShadowProb(TemplateWork &w, u32 side, const Cand &s)
{ if (w.nShadowProbs[side] < w.caps.prob) w.shadowProbs[side][w.nShadowProbs[side]++] = makeShadowProb(s); else w.overflow = 1; }

// acc + t[0] + t[1] + ... in exactly that order.  Eight terms are fetched before they are added: the chain of dependent additions then
// does not wait for loads.
ISAAC_HD double addInOrder(double acc, const double *t, u32 n)
{
    u32 i = 0;
    if (n >= 8)
    {   // the next eight terms are on their way while the current eight are added
        double a0 = t[0], a1 = t[1], a2 = t[2], a3 = t[3], a4 = t[4], a5 = t[5], a6 = t[6], a7 = t[7];
        for (i = 8; i + 8 <= n; i += 8)
        {
            const double b0 = t[i], b1 = t[i + 1], b2 = t[i + 2], b3 = t[i + 3], b4 = t[i + 4], b5 = t[i + 5], b6 = t[i + 6], b7 = t[i + 7];
            acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        }
        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
    }
    for (; i < n; ++i) acc += t[i];
    return acc;
}

// sumUniqueShadowProbabilities / sumUniquePairProbabilities (TemplateBuilder.cpp:694-714): std::sort, then the first element of
// every run of elements equal to it (std::unique_copy over forward iterators), summed in sorted order
//
// The reference's comparators use |a - b| <= 1e-7 for the probabilities, so they are strict weak orders only on data without
// "near ties" (same position, probabilities different but within 1e-7).  Without near ties the sorted sequence is unique up to
// the order of elements with identical keys, and those contribute the same term whichever of them std::unique_copy keeps: any
// correct sort gives the reference's sum.  The fast path therefore sorts by a total order, checks for near ties (adjacent
// elements suffice once sorted) and only falls back to the instruction-exact std::sort replica if it finds one.
static const u32 FAST_SORT_MIN = 192;
// Lists of up to four entries (the usual case: one or two candidates per read and what they rescued) never touch the work
// arrays: the entries are fetched side by side, ordered by the same total order in registers, and summed in that order; a
// near tie sends the list through the exact sort like any other.
static const u32 SMALL_SUM_MAX = 4;
struct ShadowKey { u64 pos; double lp; i64 obs; u32 idx; };
struct PairKey { u64 pos1, pos2; double lp; i64 obs1, obs2; u32 idx; };
ISAAC_HD bool keyLess(const ShadowKey &a, const ShadowKey &b)
{ if (a.pos != b.pos) return a.pos < b.pos; if (a.lp != b.lp) return a.lp < b.lp; if (a.obs != b.obs) return a.obs < b.obs; return a.idx < b.idx; }
ISAAC_HD bool keyNearTie(const ShadowKey &a, const ShadowKey &b) { return a.pos == b.pos && a.lp != b.lp && lpEquals(a.lp, b.lp); }
ISAAC_HD bool keyEqual(const ShadowKey &a, const ShadowKey &b) { return a.pos == b.pos && lpEquals(a.lp, b.lp) && a.obs == b.obs; }
ISAAC_HD bool keyLess(const PairKey &a, const PairKey &b)
{
    if (a.pos1 != b.pos1) return a.pos1 < b.pos1;
    if (a.pos2 != b.pos2) return a.pos2 < b.pos2;
    if (a.lp != b.lp) return b.lp < a.lp;                                // higher probability first
    if (a.obs1 != b.obs1) return a.obs1 < b.obs1;
    if (a.obs2 != b.obs2) return a.obs2 < b.obs2;
    return a.idx < b.idx;
}
ISAAC_HD bool keyNearTie(const PairKey &a, const PairKey &b) { return a.pos1 == b.pos1 && a.pos2 == b.pos2 && a.lp != b.lp && lpEquals(a.lp, b.lp); }
ISAAC_HD bool keyEqual(const PairKey &a, const PairKey &b) { return a.pos1 == b.pos1 && a.pos2 == b.pos2 && lpEquals(a.lp, b.lp) && a.obs1 == b.obs1 && a.obs2 == b.obs2; }
template <typename K> ISAAC_HD void keyOrder(K &a, K &b) { if (keyLess(b, a)) { const K t = a; a = b; b = t; } }
// k0..k3: the list's n entries, the rest padded with keys that sort last.  false: a near tie, nothing decided.
template <typename K> ISAAC_HD bool smallUniqueSum(K k0, K k1, K k2, K k3, u32 n, double &ret)
{
    keyOrder(k0, k1); keyOrder(k2, k3); keyOrder(k0, k2); keyOrder(k1, k3); keyOrder(k1, k2);
    if ((1 < n && keyNearTie(k0, k1)) || (2 < n && keyNearTie(k1, k2)) || (3 < n && keyNearTie(k2, k3))) return false;
    ret = 0.0;
    K prev = k0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (u32 i = 0; i < n; ++i)
    {
        const K cur = k0; k0 = k1; k1 = k2; k2 = k3;
        if (!(i && keyEqual(prev, cur))) ret += exp(cur.lp);
        prev = cur;
    }
    return true;
}
// the same for up to eight entries (19-comparator network); used for the shadow lists, whose keys are small enough
static const u32 SMALL_SHADOW_SUM_MAX = 8;
template <typename K> ISAAC_HD bool smallUniqueSum8(K k0, K k1, K k2, K k3, K k4, K k5, K k6, K k7, u32 n, double &ret)
{
    keyOrder(k0, k1); keyOrder(k2, k3); keyOrder(k4, k5); keyOrder(k6, k7);
    keyOrder(k0, k2); keyOrder(k1, k3); keyOrder(k4, k6); keyOrder(k5, k7);
    keyOrder(k1, k2); keyOrder(k5, k6); keyOrder(k0, k4); keyOrder(k3, k7);
    keyOrder(k1, k5); keyOrder(k2, k6);
    keyOrder(k1, k4); keyOrder(k3, k6);
    keyOrder(k2, k4); keyOrder(k3, k5);
    keyOrder(k3, k4);
    if ((1 < n && keyNearTie(k0, k1)) || (2 < n && keyNearTie(k1, k2)) || (3 < n && keyNearTie(k2, k3)) || (4 < n && keyNearTie(k3, k4)) ||
        (5 < n && keyNearTie(k4, k5)) || (6 < n && keyNearTie(k5, k6)) || (7 < n && keyNearTie(k6, k7))) return false;
    ret = 0.0;
    K prev = k0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (u32 i = 0; i < n; ++i)
    {
        const K cur = k0; k0 = k1; k1 = k2; k2 = k3; k3 = k4; k4 = k5; k5 = k6; k6 = k7;
        if (!(i && keyEqual(prev, cur))) ret += exp(cur.lp);
        prev = cur;
    }
    return true;
}
ISAAC_HD ShadowKey shadowKey(const ShadowProb *v, u32 i, u32 n)
{
    ShadowKey k; k.idx = i;
    if (i < n) { const ShadowProb p = v[i]; k.pos = p.pos; k.lp = p.logProbability; k.obs = p.observedLength; } else { k.pos = ~u64(0); k.lp = 0.0; k.obs = 0; }
    return k;
}
ISAAC_HD PairKey pairKey(const PairProb *v, u32 i, u32 n)
{
    PairKey k; k.idx = i;
    if (i < n) { const PairProb p = v[i]; k.pos1 = p.r1.pos; k.pos2 = p.r2.pos; k.lp = pairLp(p); k.obs1 = p.r1.observedLength; k.obs2 = p.r2.observedLength; }
    else { k.pos1 = ~u64(0); k.pos2 = ~u64(0); k.lp = 0.0; k.obs1 = 0; k.obs2 = 0; }
    return k;
}
ISAAC_HD double sumUniqueShadowProbabilities(TemplateCtx &x, u32 side)
{
    if (RESCUE_PRECOMPUTED == x.rescueMode) return x.sums->shadow[side];
    TemplateWork &w = *x.w;
    const u32 n = w.nShadowProbs[side]; const ShadowProb *v = w.shadowProbs[side];
    if (n <= SMALL_SUM_MAX)
    {
        double sum;
        if (smallUniqueSum(shadowKey(v, 0, n), shadowKey(v, 1, n), shadowKey(v, 2, n), shadowKey(v, 3, n), n, sum)) return sum;
    }
    else if (n <= SMALL_SHADOW_SUM_MAX && 1 == x.lanes)
    {
        double sum;
        if (smallUniqueSum8(shadowKey(v, 0, n), shadowKey(v, 1, n), shadowKey(v, 2, n), shadowKey(v, 3, n), shadowKey(v, 4, n), shadowKey(v, 5, n), shadowKey(v, 6, n), shadowKey(v, 7, n), n, sum)) return sum;
    }
    bool exact = true;
    ISAAC_PROF_T0(x);
    if (x.fastSort && n >= FAST_SORT_MIN)
    {
        for (u32 i = x.lane; i < n; i += x.lanes) w.sortIdx[i] = u16(i);
        coopSync(x);
        ShadowProbTotalLess tl; tl.v = v;
        sortTotal(x, w.sortIdx, n, tl);
        bool nearTie = false;
        for (u32 i = 1 + x.lane; i < n; i += x.lanes)
        {
            const ShadowProb &a = v[w.sortIdx[i - 1]], &b = v[w.sortIdx[i]];
            if (a.pos == b.pos && a.logProbability != b.logProbability && lpEquals(a.logProbability, b.logProbability))
            {
                nearTie = true;
#if defined(ISAAC_DEBUG_NEARTIE) && !defined(__HIP_DEVICE_COMPILE__)
                printf("near tie cluster %u n %u: pos %llu lp %.17g %.17g obs %lld %lld\n", x.clusterId, n, (unsigned long long)a.pos, a.logProbability, b.logProbability, (long long)a.observedLength, (long long)b.observedLength);
#endif
            }
        }
        exact = coopAny(x, nearTie);
    }
    if (exact)
    {
        for (u32 i = 0; i < n; ++i) w.sortIdx[i] = u16(i);
        ShadowProbIdxLess less; less.v = v;
        exactSort(w.sortIdx, i32(n), less);
    }
    ISAAC_PROF_ADD(x, 3);
    double ret = 0.0;
    if (!exact)
    {   // without near ties "equal to the first element of the run" is "equal to the predecessor": every element decides alone
        // whether it contributes, the exps run side by side and only the additions stay in sequence (x + 0.0 == x)
        for (u32 i = x.lane; i < n; i += x.lanes)
            w.terms[i] = (i && shadowProbEqual(v[w.sortIdx[i - 1]], v[w.sortIdx[i]])) ? 0.0 : exp(v[w.sortIdx[i]].logProbability);
        coopSync(x);
        ret = addInOrder(ret, w.terms, n);
    }
    else for (u32 i = 0; i < n;)
    {
        ret += exp(v[w.sortIdx[i]].logProbability);
        u32 j = i + 1;
        while (j < n && shadowProbEqual(v[w.sortIdx[i]], v[w.sortIdx[j]])) ++j;
        i = j;
    }
    ISAAC_PROF_ADD(x, 5);
    return ret;
}
ISAAC_HD double sumUniquePairProbabilities(TemplateCtx &x)
{
    if (RESCUE_PRECOMPUTED == x.rescueMode) return x.sums->pair;
    TemplateWork &w = *x.w;
    const u32 n = w.nPairProbs; const PairProb *v = w.pairProbs;
    if (n <= SMALL_SUM_MAX)
    {
        double sum;
        if (smallUniqueSum(pairKey(v, 0, n), pairKey(v, 1, n), pairKey(v, 2, n), pairKey(v, 3, n), n, sum)) return sum;
    }

    bool exact = true;
    ISAAC_PROF_T0(x);
    if (x.fastSort && n >= FAST_SORT_MIN)
    {
        for (u32 i = x.lane; i < n; i += x.lanes) w.sortIdx[i] = u16(i);
        coopSync(x);
        PairProbTotalLess tl; tl.v = v;
        sortTotal(x, w.sortIdx, n, tl);
        bool nearTie = false;
        for (u32 i = 1 + x.lane; i < n; i += x.lanes)
        {
            const PairProb &a = v[w.sortIdx[i - 1]], &b = v[w.sortIdx[i]];
            const double la = pairLp(a), lb = pairLp(b);
            if (a.r1.pos == b.r1.pos && a.r2.pos == b.r2.pos && la != lb && lpEquals(la, lb)) nearTie = true;
        }
        exact = coopAny(x, nearTie);
    }
    if (exact)
    {
        for (u32 i = 0; i < n; ++i) w.sortIdx[i] = u16(i);
        PairProbIdxLess less; less.v = v;
        exactSort(w.sortIdx, i32(n), less);
    }
    ISAAC_PROF_ADD(x, 4);
    double ret = 0.0;
    if (!exact)
    {
        for (u32 i = x.lane; i < n; i += x.lanes)
            w.terms[i] = (i && pairProbEqual(v[w.sortIdx[i - 1]], v[w.sortIdx[i]])) ? 0.0 : exp(pairLp(v[w.sortIdx[i]]));
        coopSync(x);
        ret = addInOrder(ret, w.terms, n);
    }
    else for (u32 i = 0; i < n;)
    {
        ret += exp(pairLp(v[w.sortIdx[i]]));
        u32 j = i + 1;
        while (j < n && pairProbEqual(v[w.sortIdx[i]], v[w.sortIdx[j]])) ++j;
        i = j;
    }
    return ret;
}

// TemplateBuilder::rescueShadow (TemplateBuilder.cpp:495-676): exactly one read has candidates
ISAAC_HD bool templateRescueShadow(TemplateCtx &x, BamTemplate &t, double logMismatchQ40)
{
    TemplateWork &w = *x.w;
    const u32 orphanIndex = x.nCands[0] ? 0 : 1;
    const u32 shadowIndex = (orphanIndex + 1) % 2;
    const Cand *orphans = x.cands[orphanIndex]; const u32 nOrphans = x.nCands[orphanIndex];
    const u32 bestOrphanIt = getBestFragment(x, orphanIndex);
    BestPairInfo &bestPair = w.bestRescued;
    bestPair.clear();
    bestPair.push(orphanIndex, bestOrphanIt);
    w.nShadowProbs[orphanIndex] = 0;
    for (u32 oi = 0; oi < nOrphans; ++oi)
    {
        const Cand &orphan = orphans[oi];
        w.nShadows = 0;
        if (lpLess(orphan.logProbability + 100.0, orphans[bestOrphanIt].logProbability)) { }
        else if (shadowRescue(x, orphan, 0))
        {
            const Cand &bestRescued = *x.bestRescued;
            const double currentTemplateLogProbability = orphan.logProbability + bestRescued.logProbability;
            const u64 templateScore = u64(orphan.smithWatermanScore + bestRescued.smithWatermanScore);
            if (!isVeryBadAlignment(bestRescued, x.bestRescuedPool, logMismatchQ40))
            {
                if (0 == bestPair.resolvedTemplateCount || templateScore < bestPair.bestTemplateScore ||
                    (templateScore == bestPair.bestTemplateScore && lpLess(bestPair.bestTemplateLogProbability, currentTemplateLogProbability)))
                {
                    bestPair.bestTemplateLogProbability = currentTemplateLogProbability; bestPair.bestTemplateScore = templateScore;
                    bestPair.n[orphanIndex] = 0; bestPair.push(orphanIndex, oi);
                    w.nBestOrphanShadows[orphanIndex] = 0;
                    w.bestOrphanShadows[orphanIndex][w.nBestOrphanShadows[orphanIndex]++] = cloneWithCigar(w, bestRescued, x.bestRescuedPool);
                }
                else if (templateScore == bestPair.bestTemplateScore && lpEquals(currentTemplateLogProbability, bestPair.bestTemplateLogProbability))
                {
                    bestPair.push(orphanIndex, oi);
                    if (w.nBestOrphanShadows[orphanIndex] < w.caps.best) w.bestOrphanShadows[orphanIndex][w.nBestOrphanShadows[orphanIndex]++] = cloneWithCigar(w, bestRescued, x.bestRescuedPool);
                    else w.overflow = 1;
                }
                ++bestPair.resolvedTemplateCount;
            }
        }
        if (RESCUE_PRECOMPUTED != x.rescueMode)
        {
            const u32 probBase = w.nShadowProbs[orphanIndex], probRoom = w.caps.prob - probBase;
            for (u32 s = x.lane; s < w.nShadows; s += x.lanes) if (s < probRoom) w.shadowProbs[orphanIndex][probBase + s] = makeShadowProb(w.shadowList[s]);
            coopSync(x);
            if (w.nShadows > probRoom) w.overflow = 1;
            w.nShadowProbs[orphanIndex] = probBase + imin(w.nShadows, probRoom);
            // a running fp64 sum in list order: the order of the additions is part of the result
            // (the terms side by side when the term array has room for them, the additions one after the other)
            if (w.nShadows <= imax(imax(w.caps.pos, w.caps.prob), w.caps.pair))
            {
                for (u32 s = x.lane; s < w.nShadows; s += x.lanes) w.terms[s] = exp(orphan.logProbability + w.shadowList[s].logProbability);
                coopSync(x);
                bestPair.totalTemplateProbability = addInOrder(bestPair.totalTemplateProbability, w.terms, w.nShadows);
                coopSync(x);
            }
            else for (u32 s = 0; s < w.nShadows; ++s) bestPair.totalTemplateProbability += exp(orphan.logProbability + w.shadowList[s].logProbability);
        }
    }
    if (RESCUE_PRECOMPUTED == x.rescueMode) bestPair.totalTemplateProbability = x.sums->ordered;      // the loop's running sum, made by k_cluster_sums
    if (bestPair.overflow) w.overflow = 1;
    const double totalShadowProbability = (0 < bestPair.resolvedTemplateCount) ? sumUniqueShadowProbabilities(x, orphanIndex) : 0.0;
    bool ret = true;
    Frag &orphanF = t.f[orphanIndex]; Frag &shadowF = t.f[shadowIndex];
    if (0 < bestPair.resolvedTemplateCount)
    {
        const u32 repeatIndex = x.P->scatterRepeats ? x.clusterId % bestPair.n[orphanIndex] : 0;
        const u32 listIdx = bestPair.frags[orphanIndex][repeatIndex];
        fragFromList(x, orphanF, orphanIndex, listIdx);
        Cand &orphan = orphanF.c;
        Cand &bestShadow = w.bestOrphanShadows[orphanIndex][repeatIndex];
        const bool assumeWellAnchored = updateMappingScore(x, orphan, orphanIndex, listIdx, 0 == u32(orphan.editDistance) + u32(bestShadow.editDistance));
        if (assumeWellAnchored)
        {
            const double shadowRog = x.rogRead[bestShadow.readIndex];
            const double otherShadowsProbability = (totalShadowProbability - exp(bestShadow.logProbability)) + shadowRog;
            bestShadow.alignmentScore = mapqFloor(x, otherShadowsProbability / (totalShadowProbability + shadowRog));
            const double otherPairsProbability = (bestPair.totalTemplateProbability - exp(bestPair.bestTemplateLogProbability)) + x.rog;
            t.alignmentScore = mapqFloor(x, otherPairsProbability / (bestPair.totalTemplateProbability + x.rog));
            if (!orphan.alignmentScore || !candWellAnchored(orphan))
            {
                t.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, t.alignmentScore);
                bestShadow.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bestShadow.alignmentScore);
                orphan.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.alignmentScore);
            }
        }
        else ret = flagDodgyTemplate2(x, orphan, bestShadow, t);
        shadowF.c = bestShadow; shadowF.pool = w.templateCigar;
        t.properPair = TLS_NOMINAL == tlsCheckModel(*x.tls, orphan, bestShadow);
    }
    else
    {
        fragFromList(x, orphanF, orphanIndex, bestOrphanIt);
        Cand &orphan = orphanF.c; Cand &shadow = shadowF.c;
        if (isVeryBadAlignment(orphan, orphanF.pool, logMismatchQ40)) { candSetNoMatch(orphan); candSetNoMatch(shadow); ret = false; }
        else
        {
            shadow.contigId = orphan.contigId; shadow.position = orphan.position; shadow.readIndex = u8(shadowIndex); shadow.alignmentScore = 0; shadow.cigarLength = 0;
            if (!updateMappingScore(x, orphan, orphanIndex, bestOrphanIt, 0 == orphan.editDistance)) ret = flagDodgyTemplate2(x, orphan, shadow, t);
            else
            {
                if (!candWellAnchored(orphan)) orphan.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.alignmentScore);
                t.alignmentScore = 0;
            }
        }
    }
    return ret;
}

// BestPairInfo::getBestTemplateLength (TemplateBuilder.hh:276-287)
ISAAC_HD i64 getBestTemplateLength(const TemplateCtx &x, const BestPairInfo &b)
{
    if (!b.resolvedTemplateCount) return 0;
    const Cand &a = x.cands[0][b.frags[0][0]], &c = x.cands[1][b.frags[1][0]];
    const u64 templateStart = imin(candFStrandPos(a), candFStrandPos(c));
    const u64 templateEnd = imax(candRStrandPos(a), candRStrandPos(c));
    return i64(refposPosition(templateEnd)) - i64(refposPosition(templateStart));
}

// scoreDisjoinedTemplate (TemplateBuilder.cpp:868-1008)
ISAAC_HD bool scoreDisjoinedTemplate(TemplateCtx &x, BamTemplate &t, const BestPairInfo &bestOrphans, const BestPairInfo &knownBestPair, u32 bestOrphanIndex,
                                     double totalShadowProbability, double totalOrphanProbability, const u32 bestDisjoinedFragments[2])
{
    TemplateWork &w = *x.w;
    bool ret = true;
    if (0 < bestOrphans.resolvedTemplateCount)
    {
        const u32 repeatIndex = x.P->scatterRepeats ? x.clusterId % bestOrphans.n[bestOrphanIndex] : 0;
        const u32 orphanListIdx = bestOrphans.frags[bestOrphanIndex][repeatIndex];
        const Cand &bestOrphan = x.cands[bestOrphanIndex][orphanListIdx];
        Cand &bestShadow = w.bestOrphanShadows[bestOrphanIndex][repeatIndex];
        const u32 orphanRead = bestOrphan.readIndex, shadowRead = bestShadow.readIndex;
        const bool rediscovered = !repeatIndex && knownBestPair.resolvedTemplateCount &&
            candEqual(x.cands[orphanRead][knownBestPair.frags[orphanRead][0]], bestOrphan) &&
            candEqual(x.cands[shadowRead][knownBestPair.frags[shadowRead][0]], bestShadow);
        Frag &orphanF = t.f[orphanRead];
        fragFromList(x, orphanF, bestOrphanIndex, orphanListIdx);
        Cand &orphan = orphanF.c;
        const bool shadowWellAnchored = rediscovered && candWellAnchored(x.cands[shadowRead][knownBestPair.frags[shadowRead][0]]);
        const bool assumeWellAnchored = updateMappingScore(x, orphan, orphanRead, bestOrphans.frags[orphanRead][repeatIndex],
                                                           0 == u32(orphan.editDistance) + u32(bestShadow.editDistance) || shadowWellAnchored);
        t.properPair = TLS_NOMINAL == tlsCheckModel(*x.tls, orphan, bestShadow);
        if (assumeWellAnchored)
        {
            const double shadowRog = x.rogRead[shadowRead];
            const double otherShadowsProbability = (totalShadowProbability - exp(bestShadow.logProbability)) + shadowRog;
            bestShadow.alignmentScore = mapqFloor(x, otherShadowsProbability / (totalShadowProbability + shadowRog));
            const double orphanRog = x.rogRead[orphanRead];
            const double otherOrphansProbability = (totalOrphanProbability - exp(bestOrphan.logProbability)) + orphanRog;
            orphan.alignmentScore = mapqFloor(x, otherOrphansProbability / (totalOrphanProbability + orphanRog));
            const double otherPairsProbability = (bestOrphans.totalTemplateProbability - exp(bestOrphans.bestTemplateLogProbability)) + x.rog;
            t.alignmentScore = mapqFloor(x, otherPairsProbability / (bestOrphans.totalTemplateProbability + x.rog));
            if ((!orphan.alignmentScore || !candWellAnchored(orphan)) && (!bestShadow.alignmentScore || !shadowWellAnchored))
            {
                t.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, t.alignmentScore);
                bestShadow.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bestShadow.alignmentScore);
                orphan.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.alignmentScore);
            }
            t.f[shadowRead].c = bestShadow; t.f[shadowRead].pool = w.templateCigar;
        }
        else
        {
            ret = flagDodgyTemplate2(x, orphan, bestShadow, t);
            t.f[shadowRead].c = bestShadow; t.f[shadowRead].pool = w.templateCigar;
        }
    }
    else if (knownBestPair.resolvedTemplateCount) ret = flagDodgyTemplate2(x, t.f[0].c, t.f[1].c, t);
    else
    {
        fragFromList(x, t.f[0], 0, bestDisjoinedFragments[0]);
        fragFromList(x, t.f[1], 1, bestDisjoinedFragments[1]);
        Cand &read1 = t.f[0].c, &read2 = t.f[1].c;
        t.alignmentScore = 0; t.properPair = false;
        const bool a1 = updateMappingScore(x, read1, 0, bestDisjoinedFragments[0], 0 == read1.editDistance);
        const bool a2 = updateMappingScore(x, read2, 1, bestDisjoinedFragments[1], 0 == read2.editDistance);
        if (!a1 && !a2) ret = flagDodgyTemplate2(x, read1, read2, t);
        else
        {
            if (!candWellAnchored(read1)) read1.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, read1.alignmentScore);
            if (!candWellAnchored(read2)) read2.alignmentScore = imin(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, read2.alignmentScore);
        }
    }
    return ret;
}

// buildDisjoinedTemplate (TemplateBuilder.cpp:716-866)
ISAAC_HD bool buildDisjoinedTemplate(TemplateCtx &x, BamTemplate &t, const BestPairInfo &knownBestPair, double logMismatchQ40)
{
    TemplateWork &w = *x.w;
    const u32 bestDisjoinedFragments[2] = { getBestFragment(x, 0), getBestFragment(x, 1) };
    const i64 bestTemplateLength = getBestTemplateLength(x, knownBestPair);
    u32 bestOrphanIndex = 0;
    BestPairInfo &bestOrphans = w.bestRescued;
    bestOrphans.init(bestDisjoinedFragments[0], bestDisjoinedFragments[1]);
    w.nPairProbs = 0;
    for (u32 orphanIndex = 0; 2 > orphanIndex; ++orphanIndex)
    {
        w.nShadowProbs[orphanIndex] = 0;
        w.nBestOrphanShadows[orphanIndex] = 0;
        const Cand *orphans = x.cands[orphanIndex]; const u32 nOrphans = x.nCands[orphanIndex];
        for (u32 oi = 0; oi < nOrphans; ++oi)
        {
            const Cand &orphan = orphans[oi];
            const bool skipThisOrphan = knownBestPair.resolvedTemplateCount ?
                u32(orphan.editDistance) > (knownBestPair.bestPairEditDistance + SKIP_ORPHAN_EDIT_DISTANCE) :
                lpLess(orphan.logProbability + 100.0, orphans[bestDisjoinedFragments[orphanIndex]].logProbability);
            w.nShadows = 0;
            STAMP_BEGIN();
            const bool rescued = !skipThisOrphan && shadowRescue(x, orphan, bestTemplateLength);
            STAMP(39);
            if (rescued)
            {
                const Cand &bestRescued = *x.bestRescued;
                const double currentTemplateLogProbability = orphan.logProbability + bestRescued.logProbability;
                const u32 rescuedEditDistance = u32(orphan.editDistance) + u32(bestRescued.editDistance);
                if (isVeryBadAlignment(bestRescued, x.bestRescuedPool, logMismatchQ40)) { }
                else if (!knownBestPair.resolvedTemplateCount || (knownBestPair.bestPairEditDistance + SKIP_ORPHAN_EDIT_DISTANCE) >= rescuedEditDistance)
                {
                    const u64 templateScore = u64(orphan.smithWatermanScore + bestRescued.smithWatermanScore);
                    if (0 == bestOrphans.resolvedTemplateCount || templateScore < bestOrphans.bestTemplateScore ||
                        (templateScore == bestOrphans.bestTemplateScore && lpLess(bestOrphans.bestTemplateLogProbability, currentTemplateLogProbability)))
                    {
                        bestOrphans.bestTemplateLogProbability = currentTemplateLogProbability; bestOrphans.bestTemplateScore = templateScore;
                        bestOrphans.n[orphanIndex] = 0; bestOrphans.push(orphanIndex, oi);
                        w.nBestOrphanShadows[orphanIndex] = 0;
                        w.bestOrphanShadows[orphanIndex][w.nBestOrphanShadows[orphanIndex]++] = cloneWithCigar(w, bestRescued, x.bestRescuedPool);
                        bestOrphanIndex = orphanIndex;
                    }
                    else if (templateScore == bestOrphans.bestTemplateScore && lpEquals(currentTemplateLogProbability, bestOrphans.bestTemplateLogProbability))
                    {
                        bestOrphans.push(orphanIndex, oi);
                        if (w.nBestOrphanShadows[orphanIndex] < w.caps.best) w.bestOrphanShadows[orphanIndex][w.nBestOrphanShadows[orphanIndex]++] = cloneWithCigar(w, bestRescued, x.bestRescuedPool);
                        else w.overflow = 1;
                    }
                    ++bestOrphans.resolvedTemplateCount;
                }
            }
            ISAAC_PROF_T0(x);
            if (RESCUE_PRECOMPUTED != x.rescueMode)
            {   // every shadow contributes one pair and one shadow probability entry; the entries are independent of each other
                const u32 pairBase = w.nPairProbs, pairRoom = w.caps.pair - pairBase;
                const u32 probBase = w.nShadowProbs[orphanIndex], probRoom = w.caps.prob - probBase;
                for (u32 s = x.lane; s < w.nShadows; s += x.lanes)
                {
                    const Cand &shadow = w.shadowList[s];
                    if (s < pairRoom)
                    {
                        PairProb &pp = w.pairProbs[pairBase + s];
                        pp.r1 = makeShadowProb(0 == orphanIndex ? orphan : shadow); pp.r2 = makeShadowProb(0 == orphanIndex ? shadow : orphan);
                    }
                    if (s < probRoom) w.shadowProbs[orphanIndex][probBase + s] = makeShadowProb(shadow);
                }
                coopSync(x);
                if (w.nShadows > pairRoom || w.nShadows > probRoom) w.overflow = 1;
                w.nPairProbs = pairBase + imin(w.nShadows, pairRoom);
                w.nShadowProbs[orphanIndex] = probBase + imin(w.nShadows, probRoom);
            }
            ISAAC_PROF_ADD(x, 2);
        }
    }
    const u32 bestShadowIndex = (bestOrphanIndex + 1) % 2;
    double totalShadowProbability = 0.0, totalOrphanProbability = 0.0;
    STAMP_BEGIN();
    if (0 < bestOrphans.resolvedTemplateCount)
    {
        if (RESCUE_PRECOMPUTED != x.rescueMode) for (u32 i = 0; i < x.nCands[bestShadowIndex]; ++i) pushShadowProb(w, bestOrphanIndex, x.cands[bestShadowIndex][i]);
        totalShadowProbability = sumUniqueShadowProbabilities(x, bestOrphanIndex);
        if (RESCUE_PRECOMPUTED != x.rescueMode) for (u32 i = 0; i < x.nCands[bestOrphanIndex]; ++i) pushShadowProb(w, bestShadowIndex, x.cands[bestOrphanIndex][i]);
        totalOrphanProbability = sumUniqueShadowProbabilities(x, bestShadowIndex);
        bestOrphans.totalTemplateProbability += sumUniquePairProbabilities(x);
    }
    if (bestOrphans.overflow) w.overflow = 1;
    STAMP(40);
    const bool scored = scoreDisjoinedTemplate(x, t, bestOrphans, knownBestPair, bestOrphanIndex, totalShadowProbability, totalOrphanProbability, bestDisjoinedFragments);
    STAMP(41);
    return scored;
}

// pickBestFragment (TemplateBuilder.cpp:1035-1058): single-ended data
ISAAC_HD bool pickBestFragment(TemplateCtx &x, BamTemplate &t)
{
    if (!x.nCands[0]) return false;
    const u32 best = getBestFragment(x, 0);
    fragFromList(x, t.f[0], 0, best);
    if (!updateMappingScore(x, t.f[0].c, 0, best, false))
    {
        if (-1 == x.P->dodgyAlignmentScore) { candSetNoMatch(t.f[0].c); t.alignmentScore = 0xffffffffu; return false; }
        t.f[0].c.alignmentScore = 0xffffffffu; t.alignmentScore = 0xffffffffu;
    }
    return true;
}

// BamTemplate::filterLowQualityFragments (BamTemplate.cpp:47-72)
ISAAC_HD bool filterLowQualityFragments(BamTemplate &t, u32 mapqThreshold)
{
    bool ret = false; u32 alignmentScore = 0;
    for (u32 i = 0; t.n > i; ++i)
    {
        Cand &fragment = t.f[i].c;
        if (mapqThreshold > fragment.alignmentScore)
        {
            fragment.cigarLength = 0; fragment.cigarOffset = 0; fragment.alignmentScore = 0;
            const Cand &mate = t.f[(i + 1) % t.n].c;
            fragment.position = mate.position; fragment.contigId = mate.contigId;
        }
        else if (candAligned(fragment)) ret = true;
        alignmentScore += fragment.alignmentScore;
    }
    t.alignmentScore = alignmentScore;
    return ret;
}

ISAAC_HD void bamTemplateInitialize(const TemplateCtx &x, BamTemplate &t)
{
    t.n = x.P->nReads; t.alignmentScore = 0; t.properPair = false;
    for (u32 i = 0; i < 2; ++i) { candInit(t.f[i].c, i); t.f[i].pool = x.frags->cigarPool; }
}

// TemplateBuilder::buildTemplate (TemplateBuilder.cpp:97-174) incl. pickBestPair (:1060-1086)
ISAAC_HD bool buildTemplate(TemplateCtx &x, BamTemplate &t, double logMismatchQ40)
{
    TemplateWork &w = *x.w;
    w.templateCigarUsed = 0;
    bamTemplateInitialize(x, t);
    bool ret;
    const u32 n0 = x.nCands[0], n1 = x.nCands[1];
    if (2 == x.P->nReads)
    {
        if (n0 && n1)
        {
            STAMP_BEGIN();
            locateBestPair(x, w.bestCombination);
            STAMP(36);
            if (w.bestCombination.overflow) w.overflow = 1;
            if (!w.bestCombination.resolvedTemplateCount || !buildPairedEndTemplate(x, t, w.bestCombination) || w.bestCombination.bestPairEditDistance)
            {
                STAMP(37);
                ret = buildDisjoinedTemplate(x, t, w.bestCombination, logMismatchQ40);
                STAMP(38);
            }
            else ret = true;
        }
        else if (n0 || n1) ret = templateRescueShadow(x, t, logMismatchQ40);
        else ret = false;
    }
    else ret = pickBestFragment(x, t);
    if (ret && 0xffffffffu != t.alignmentScore)
    {
        if (!t.properPair) ret = filterLowQualityFragments(t, x.P->mapqThreshold);
        else if (x.P->mapqThreshold > t.alignmentScore) { filterLowQualityFragments(t, 0xffffffffu); ret = false; }
    }
    return ret;
}

// ------------------------------------------------------------------------------------------------------------------
// end clippers; new CIGARs are appended to the template buffer
ISAAC_HD void tcPush(TemplateWork &w, u32 word) { if (w.templateCigarUsed < w.caps.templateCigar) w.templateCigar[w.templateCigarUsed++] = word; else w.overflow = 1; }

// clipMismatches<5> (Alignment.hh:55-88) walking forward (dir = +1) or backward (dir = -1)
ISAAC_HD void clipMismatches(const ReadView &read, bool reverse, i64 seqIdx, i64 seqCount, const char *reference, i64 refIdx, i64 refCount, i32 dir, u32 &clipped, u32 &editAdj)
{
    const u32 MIN = 5;
    u32 matchesInARow = 0, edMismatches = 0, edUnclipped = 0, ret = 0;
    while (seqCount && refCount && MIN > matchesInARow)
    {
        const char s = strandBase(read, reverse, u32(seqIdx)); const char r = reference[refIdx];
        if (isMatch(s, r)) { ++matchesInARow; edUnclipped += (s != r); }
        else { matchesInARow = 0; edUnclipped = 0; }
        edMismatches += (s != r);
        seqIdx += dir; refIdx += dir; --seqCount; --refCount; ++ret;
    }
    if (MIN == matchesInARow) { clipped = ret - matchesInARow; editAdj = edMismatches - edUnclipped; }
    else { clipped = 0; editAdj = 0; }
}

// SemialignedEndsClipper::clipLeftSide / clipRightSide (SemialignedEndsClipper.cpp:31-156)
ISAAC_HD bool semialignedClipLeft(TemplateCtx &x, Frag &fr)
{
    TemplateWork &w = *x.w; Cand &f = fr.c;
    const ReadView &read = x.reads[f.readIndex];
    u32 oldOffset = f.cigarOffset, oldLength = f.cigarLength;
    u32 op = fr.pool[oldOffset];
    u32 softClippedBeginBases = 0; i64 seqBegin = 0;
    if (OP_SOFT_CLIP == cigarCode(op))
    {
        if (2 > f.cigarLength) return false;
        ++oldOffset; --oldLength; softClippedBeginBases = cigarLen(op); seqBegin += cigarLen(op);
        op = fr.pool[oldOffset];
    }
    if (OP_ALIGN == cigarCode(op))
    {
        u32 mappedBeginBases = cigarLen(op);
        const char *reference = x.R->bases + x.R->contigOffset[f.contigId];
        const i64 refSize = i64(contigLength(*x.R, f.contigId));
        u32 clipped, editAdj;
        clipMismatches(read, f.reverse, seqBegin, mappedBeginBases, reference, f.position, refSize - f.position, +1, clipped, editAdj);
        if (clipped)
        {
            const u32 newOffset = w.templateCigarUsed;
            f.observedLength -= clipped; softClippedBeginBases += clipped; mappedBeginBases -= clipped; f.position += clipped; f.editDistance = u16(f.editDistance - editAdj);
            tcPush(w, cigarOp(softClippedBeginBases, OP_SOFT_CLIP));
            tcPush(w, cigarOp(mappedBeginBases, OP_ALIGN));
            for (u32 i = oldOffset + 1; i < oldOffset + oldLength; ++i) tcPush(w, fr.pool[i]);
            fr.pool = w.templateCigar; f.cigarOffset = newOffset; f.cigarLength = u16(w.templateCigarUsed - newOffset);
            return true;
        }
    }
    return false;
}
ISAAC_HD bool semialignedClipRight(TemplateCtx &x, Frag &fr)
{
    TemplateWork &w = *x.w; Cand &f = fr.c;
    const ReadView &read = x.reads[f.readIndex];
    const u32 oldOffset = f.cigarOffset; u32 oldLength = f.cigarLength;
    u32 op = fr.pool[oldOffset + oldLength - 1];
    u32 softClippedEndBases = 0; i64 seqR = i64(read.length) - 1;   // reverse iterator: last base first
    if (OP_SOFT_CLIP == cigarCode(op))
    {
        if (2 > f.cigarLength) return false;
        --oldLength; softClippedEndBases = cigarLen(op); seqR -= cigarLen(op);
        op = fr.pool[oldOffset + oldLength - 1];
    }
    if (OP_ALIGN == cigarCode(op))
    {
        u32 mappedEndBases = cigarLen(op);
        const char *reference = x.R->bases + x.R->contigOffset[f.contigId];
        const i64 refLast = f.position + i64(candObservedLength(f)) - 1;
        u32 clipped, editAdj;
        clipMismatches(read, f.reverse, seqR, mappedEndBases, reference, refLast, refLast + 1, -1, clipped, editAdj);
        if (clipped)
        {
            const u32 newOffset = w.templateCigarUsed;
            f.observedLength -= clipped; softClippedEndBases += clipped; f.editDistance = u16(f.editDistance - editAdj); mappedEndBases -= clipped;
            for (u32 i = oldOffset; i < oldOffset + oldLength - 1; ++i) tcPush(w, fr.pool[i]);
            tcPush(w, cigarOp(mappedEndBases, OP_ALIGN));
            tcPush(w, cigarOp(softClippedEndBases, OP_SOFT_CLIP));
            fr.pool = w.templateCigar; f.cigarOffset = newOffset; f.cigarLength = u16(w.templateCigarUsed - newOffset);
            return true;
        }
    }
    return false;
}
// SemialignedEndsClipper::clip (SemialignedEndsClipper.cpp:161-205)
ISAAC_HD void semialignedClip(TemplateCtx &x, BamTemplate &t)
{
    for (u32 k = 0; k < t.n; ++k)
    {
        Frag &fr = t.f[k];
        if (!candAligned(fr.c)) continue;
        bool changed = semialignedClipLeft(x, fr);
        if (semialignedClipRight(x, fr)) changed = true;
        if (changed && 2 == t.n)
        {
            Cand &mate = t.f[t.n - 1 - fr.c.readIndex].c;
            if (!candAligned(mate)) { mate.position = fr.c.position; break; }
        }
    }
}

// OverlappingEndsClipper::clip (OverlappingEndsClipper.cpp:46-183)
ISAAC_HD void overlappingClip(TemplateCtx &x, BamTemplate &t)
{
    TemplateWork &w = *x.w;
    if (2 != t.n) return;
    Cand &r1 = t.f[0].c, &r2 = t.f[1].c;
    if (!candAligned(r1) || !candAligned(r2) || r1.gapCount || r2.gapCount) return;
    if (r1.reverse == r2.reverse) return;                       // (:62 compares r1.contigId with itself: chimeras are not skipped)
    const u32 li = r1.position < r2.position ? 0 : 1;
    const u32 ri = r1.position <= r2.position ? 1 : 0;
    Frag &leftF = t.f[li]; Frag &rightF = t.f[ri];
    Cand &left = leftF.c; Cand &right = rightF.c;
    if (left.reverse) return;
    const i64 overlapLength = left.position + i64(candObservedLength(left)) - right.position;
    if (0 >= overlapLength) return;
    const ReadView &leftRead = x.reads[left.readIndex]; const ReadView &rightRead = x.reads[right.readIndex];
    u32 leftEndSoftClip = 0; u32 leftEndOffset = leftRead.length;
    u32 leftLastIdx = left.cigarOffset + left.cigarLength - 1;
    u32 leftLastOp = leftF.pool[leftLastIdx];
    if (OP_SOFT_CLIP == cigarCode(leftLastOp))
    {
        if (left.cigarLength < 2) return;
        leftEndOffset -= cigarLen(leftLastOp); leftEndSoftClip = cigarLen(leftLastOp); --leftLastIdx; leftLastOp = leftF.pool[leftLastIdx];
    }
    if (OP_ALIGN != cigarCode(leftLastOp)) return;             // ISAAC_ASSERT in the reference
    if (overlapLength >= i64(cigarLen(leftLastOp))) return;
    u32 rightStartOffset = 0; u32 rightFirstIdx = right.cigarOffset;
    u32 rightFirstOp = rightF.pool[rightFirstIdx];
    if (OP_SOFT_CLIP == cigarCode(rightFirstOp))
    {
        if (right.cigarLength < 2) return;
        rightStartOffset += cigarLen(rightFirstOp); ++rightFirstIdx; rightFirstOp = rightF.pool[rightFirstIdx];
    }
    if (OP_ALIGN != cigarCode(rightFirstOp)) return;
    if (overlapLength >= i64(cigarLen(rightFirstOp))) return;
    i32 diff = 0;   // left forward qualities minus right reverse qualities over the overlap
    for (i64 i = 0; i < overlapLength; ++i)
        diff += i32(strandQuality(leftRead, false, u32(leftEndOffset - overlapLength + i))) - i32(strandQuality(rightRead, true, u32(rightStartOffset + i)));
    if (0 < diff)
    {
        const char *reference = x.R->bases + x.R->contigOffset[right.contigId] + right.position;
        const u32 newOffset = w.templateCigarUsed;
        const u32 oldEnd = right.cigarOffset + right.cigarLength;
        tcPush(w, cigarOp(u32(rightStartOffset + overlapLength), OP_SOFT_CLIP));
        tcPush(w, cigarOp(u32(cigarLen(rightFirstOp) - overlapLength), OP_ALIGN));
        for (u32 i = rightFirstIdx + 1; i < oldEnd; ++i) tcPush(w, rightF.pool[i]);
        candIncrementClipLeft(right, u32(overlapLength));
        right.observedLength -= u32(overlapLength);
        u32 ed = 0; for (i64 i = 0; i < overlapLength; ++i) ed += (strandBase(rightRead, true, u32(rightStartOffset + i)) != reference[i]);
        right.editDistance = u16(right.editDistance - ed);
        rightF.pool = w.templateCigar; right.cigarOffset = newOffset; right.cigarLength = u16(w.templateCigarUsed - newOffset);
    }
    else
    {
        const char *reference = x.R->bases + x.R->contigOffset[left.contigId] + left.position + i64(candObservedLength(left)) - overlapLength;
        const u32 newOffset = w.templateCigarUsed;
        for (u32 i = left.cigarOffset; i < leftLastIdx; ++i) tcPush(w, leftF.pool[i]);
        tcPush(w, cigarOp(u32(cigarLen(leftLastOp) - overlapLength), OP_ALIGN));
        tcPush(w, cigarOp(u32(leftEndSoftClip + overlapLength), OP_SOFT_CLIP));
        candIncrementClipRight(left, u32(overlapLength));
        left.observedLength -= u32(overlapLength);
        u32 ed = 0; for (i64 i = 0; i < overlapLength; ++i) ed += (strandBase(leftRead, false, u32(leftEndOffset - overlapLength + i)) != reference[i]);
        left.editDistance = u16(left.editDistance - ed);
        leftF.pool = w.templateCigar; left.cigarOffset = newOffset; left.cigarLength = u16(w.templateCigarUsed - newOffset);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// io::FragmentHeader fields (include/io/Fragment.hh:101-246) + BAM MAPQ (FragmentAccessorBamAdapter.hh:250-265)
struct FragmentRecord
{
    u64 fStrandPosition, mateFStrandPosition; i32 bamTlen; u32 observedLength;
    u16 lowClipped, highClipped, alignmentScore, templateAlignmentScore, readLength, cigarLength, gapCount, editDistance;
    u32 flags, cigarOffset, tile, clusterId, mapq, reserved;
};
static_assert(sizeof(FragmentRecord) == 64, "FragmentRecord layout");

ISAAC_HD u64 refposLocation(u64 v) { return (v >> 1) - (u64(1) << 40); }
ISAAC_HD i32 getTlen(const Cand &fragment, const Cand &mate)
{
    if (!candAligned(fragment) || !candAligned(mate)) return 0;
    const u64 fb = candFStrandPos(fragment), fe = refpos(fragment.contigId, u64(fragment.position + i64(fragment.observedLength)));
    const u64 mb = candFStrandPos(mate), me = refpos(mate.contigId, u64(mate.position + i64(mate.observedLength)));
    const u64 distance = refposLocation(imax(fe, me)) - refposLocation(imin(fb, mb));
    const bool firstRead = 0 == fragment.readIndex;
    const i64 ret = fb < mb ? i64(distance) : (mb < fb || !firstRead) ? -i64(distance) : i64(distance);
    return i32(ret);
}

ISAAC_HD void makeFragmentRecord(const TemplateCtx &x, const BamTemplate &t, u32 i, u32 tile, FragmentRecord &r)
{
    const Cand &f = t.f[i].c;
    const u16 DODGY = 0xffff;
    if (2 == t.n)
    {
        const Cand &mate = t.f[1 - i].c;
        r.bamTlen = getTlen(f, mate);
        r.fStrandPosition = candAligned(f) ? candFStrandPos(f) : candFStrandPos(mate);
        r.templateAlignmentScore = u16(t.properPair ? t.alignmentScore : f.alignmentScore);
        r.mateFStrandPosition = candAligned(mate) ? candFStrandPos(mate) : candFStrandPos(f);
        r.flags = 1u | (u32(!candAligned(f)) << 1) | (u32(!candAligned(mate)) << 2) | (u32(f.reverse) << 3) | (u32(mate.reverse) << 4) |
                  (u32(0 == f.readIndex) << 5) | (u32(1 == f.readIndex) << 6) | (u32(t.properPair) << 8);
    }
    else
    {
        r.bamTlen = 0; r.fStrandPosition = candFStrandPos(f); r.templateAlignmentScore = u16(f.alignmentScore); r.mateFStrandPosition = REFPOS_NOMATCH;
        r.flags = (u32(!candAligned(f)) << 1) | (1u << 2) | (u32(f.reverse) << 3) | (1u << 5) | (1u << 6);
    }
    r.observedLength = candObservedLength(f);
    r.lowClipped = f.lowClipped; r.highClipped = f.highClipped; r.alignmentScore = u16(f.alignmentScore);
    r.readLength = u16(x.reads[f.readIndex].length); r.cigarLength = f.cigarLength; r.gapCount = f.gapCount; r.editDistance = f.editDistance;
    r.tile = tile; r.clusterId = x.clusterId; r.reserved = 0; r.cigarOffset = 0;
    const u32 forced = u32(x.P->dodgyAlignmentScore) & 0xff;
    if (r.flags & (1u << 8)) r.mapq = (DODGY == r.templateAlignmentScore) ? forced : imin<u32>(60u, imax(r.alignmentScore, r.templateAlignmentScore));
    else r.mapq = (DODGY == r.alignmentScore) ? forced : imin<u32>(60u, r.alignmentScore);
}

// MatchSelector::processMatchList for one cluster (MatchSelector.cpp:296-366).  Returns true when the template is stored.
ISAAC_HD bool selectCluster(TemplateCtx &x, BamTemplate &t, double logMismatchQ40)
{
    x.w->overflow = 0;
    bool store;
    if (x.frags->built)
    {
        STAMP_BEGIN();
        store = buildTemplate(x, t, logMismatchQ40) || x.P->keepUnaligned;
        STAMP(33);
        if (store)
        {
            if (x.P->clipSemialigned) semialignedClip(x, t);
            STAMP(34);
            if (x.P->clipOverlapping) overlappingClip(x, t);
            STAMP(35);
        }
    }
    else { bamTemplateInitialize(x, t); store = x.P->keepUnaligned; }
    return store;
}

// Quality::restOfGenomeCorrection (Quality.hh:87-91; genome length passes through `unsigned`) and the clamp of
// RestOfGenomeCorrection.hh:51-55,80-83.  Evaluated on the HOST once per tile (glibc exp/log) and handed to the kernel.
struct RogCorrection { double read[2]; double pair; };

} // namespace isaac
