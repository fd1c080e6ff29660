// The fragment stage's per-cluster steps for short lists -- the main pass of k_build_fragments, k_finish_candidates and
// k_finish_fragments.  aligner.h states them in general: every list is an index array in per-cluster global memory, sorted by an
// instruction-exact std::sort over keys that are read through the candidates' pointers each time they are compared.  Nearly all
// clusters have lists of a few entries, and for those the same results come from much less:
//   * std::sort of libstdc++ on up to 16 elements is its insertion sort alone (__introsort_loop does nothing below _S_threshold = 16,
//     __final_insertion_sort is then one __insertion_sort), which is a stable sort; under a comparator that is a strict weak order the
//     result is THE stable order, whatever algorithm finds it.  All comparators of this stage are lexicographic orders on integer keys.
//   * the keys of a list are read once, side by side, into a small per-thread key area (LDS in the kernels), the order is sixteen
//     4-bit indexes in one 64-bit register, and candidates are written or moved only when they really change place.
// Clusters with a longer list, or that met a capacity limit, are left to the general form (listed by the kernels, run by a second
// launch): LEAN_LIST_MAX is the boundary.
//
// Behaviour follows (paths relative to /root/reference/src/c++): lib/alignment/FragmentBuilder.cpp:82-145 (build), :147-217
// (alignFragments), :219-249 (addMatch), :279-324 (consolidateDuplicateFragments), :326-343 (getReadPosition);
// lib/alignment/fragmentBuilder/SimpleIndelAligner.cpp:443-518; include/alignment/FragmentMetadata.hh:419-448.
#pragma once
#include "aligner.h"

namespace isaac
{

static const u32 LEAN_LIST_MAX = 16;

// a permutation of up to sixteen list positions, four bits each
struct Nibbles
{
    u64 v;
    ISAAC_HD u32 get(u32 i) const { return u32(v >> (4 * i)) & 15u; }
    ISAAC_HD void set(u32 i, u32 x) { v = (v & ~(u64(15) << (4 * i))) | (u64(x) << (4 * i)); }
};
static const u64 NIBBLES_IDENTITY = 0xfedcba9876543210ull;

// stable insertion sort of order[0 .. n) by the keys of the area
ISAAC_HD void leanSortOrder(Nibbles &order, u32 n, const LeanKeyArea &k)
{
    for (u32 i = 1; i < n; ++i)
    {
        const u32 val = order.get(i);
        const u64 va = k.a[val * k.stride], vb = k.b[val * k.stride];
        u32 j = i;
        while (j > 0)
        {
            const u32 prev = order.get(j - 1);
            const u64 pa = k.a[prev * k.stride];
            if (!(va < pa || (va == pa && vb < k.b[prev * k.stride]))) break;
            order.set(j, prev); --j;
        }
        if (j != i) order.set(j, val);
    }
}
// the list order applied to the candidates in place (aligner.h: applyOrderInPlace)
ISAAC_HD void leanApplyOrder(Cand *store, Nibbles order, u32 n)
{
    for (u32 i = 0; i < n; ++i)
    {
        u32 src = order.get(i);
        while (src < i) src = order.get(src);
        if (src != i) { const Cand t = store[i]; store[i] = store[src]; store[src] = t; }
        order.set(i, src);
    }
}

// FragmentBuilder::consolidateDuplicateFragments (FragmentBuilder.cpp:279-324) on a list of up to LEAN_LIST_MAX candidates, applied in
// place: the list ends up sorted, without its unaligned members (removeUnaligned), equal neighbours merged.  Returns the new length.
ISAAC_HD u32 leanConsolidate(Cand *store, u32 n, bool removeUnaligned, const LeanKeyArea &k)
{
    if (!n) return 0;
    u32 aligned = 0;
    bool sorted = true; u64 prevA = 0, prevB = 0;
    for (u32 i = 0; i < n; ++i)
    {
        const Cand &c = store[i];
        const u64 a = leanPositionKey(c.contigId, c.position), b = (u64(c.reverse) << 32) | c.observedLength;
        k.a[i * k.stride] = a; k.b[i * k.stride] = b;
        if (candAligned(c)) aligned |= 1u << i;
        if (i && (a < prevA || (a == prevA && b <= prevB))) sorted = false;         // strictly ascending: nothing to sort, nothing to merge
        prevA = a; prevB = b;
    }
    const u32 all = (n >= 32) ? 0xffffffffu : ((1u << n) - 1);
    if (sorted && (!removeUnaligned || aligned == all)) return n;
    Nibbles order; order.v = NIBBLES_IDENTITY;
    leanSortOrder(order, n, k);
    u32 first = 0;
    while (first != n && removeUnaligned && !((aligned >> order.get(first)) & 1)) ++first;
    if (first) { for (u32 i = first; i < n; ++i) order.set(i - first, order.get(i)); n -= first; }
    if (2 <= n)
    {
        u32 last = 0;
        for (u32 current = 1; current != n; ++current)
        {
            const u32 ic = order.get(current), il = order.get(last);
            if (removeUnaligned && !((aligned >> ic) & 1)) { }
            else if (k.a[il * k.stride] == k.a[ic * k.stride] && k.b[il * k.stride] == k.b[ic * k.stride])
            {
                Cand &a = store[il]; const Cand &b = store[ic];
                a.uniqueSeedCount = u16(a.uniqueSeedCount + b.uniqueSeedCount);
                a.nonUniqueFirst = imin(a.nonUniqueFirst, b.nonUniqueFirst);
                a.nonUniqueSecond = imax(a.nonUniqueSecond, b.nonUniqueSecond);
            }
            else { ++last; if (last != current) order.set(last, ic); }
        }
        n = last + 1;
    }
    leanApplyOrder(store, order, n);
    return n;
}

// The sorted candidate keys of one read written as candidates: runs of equal (contig, position, strand) are one candidate, the first of the
// run with the seeds of all of them (FragmentBuilder::consolidateDuplicateFragments on freshly added matches).  index(at): the key entry at
// list position `at`.  Key words: a = leanPositionKey, b = strand << 45 | seed position << 5 | seed << 1 | "the table marks the k-mer as having neighbours".
struct NibbleIndex { Nibbles order; ISAAC_HD u32 operator()(u32 at) const { return order.get(at); } };
struct ByteIndex { const u8 *order; ISAAC_HD u32 operator()(u32 at) const { return order[at]; } };
template <typename IndexF>
ISAAC_HD u32 leanEmitCandidates(const DevParams &P, Cand *store, u32 r, u32 repeatSeedsCount, u32 n, const LeanKeyArea &k, IndexF index)
{
    u32 stored = 0;
    for (u32 at = 0; at < n;)
    {
        const u32 i = index(at);
        const u64 a = k.a[i * k.stride], b = k.b[i * k.stride];
        const u32 s = u32(b >> 1) & 0xfu;
        const u16 offset = P.seeds[s].offset;
        u32 uniqueSeedCount = (b & 1) ? 0u : 1u;
        u16 nonUniqueFirst = (b & 1) ? offset : NON_UNIQUE_NONE, nonUniqueSecond = (b & 1) ? offset : u16(0);
        u32 next = at + 1;
        for (; next < n; ++next)
        {
            const u32 j = index(next);
            const u64 bj = k.b[j * k.stride];
            if (k.a[j * k.stride] != a || (((bj ^ b) >> 45) & 1)) break;
            const u16 offsetJ = P.seeds[u32(bj >> 1) & 0xfu].offset;
            if (bj & 1) { nonUniqueFirst = imin(nonUniqueFirst, offsetJ); nonUniqueSecond = imax(nonUniqueSecond, offsetJ); } else ++uniqueSeedCount;
        }
        Cand f;
        candInit(f, r);
        f.firstSeedIndex = (signed char)s;
        f.contigId = u32(a >> 41);
        f.position = i64(a & ((u64(1) << 41) - 1)) - LEAN_POSITION_BIAS;
        f.reverse = u8((b >> 45) & 1);
        f.repeatSeedsCount = u16(repeatSeedsCount);
        f.uniqueSeedCount = u16(uniqueSeedCount); f.nonUniqueFirst = nonUniqueFirst; f.nonUniqueSecond = nonUniqueSecond;
        store[stored++] = f;
        at = next;
    }
    return stored;
}

// ------------------------------------------------------------------------------------------------------------------
// k_build_fragments: FragmentBuilder::build (aligner.h: buildCandidates) for a cluster with up to LEAN_LIST_MAX matches whose slots
// hold them all.  The candidates are sorted as keys -- (contig, position) and (strand, seed position, seed) is the order the
// reference's match order followed by its stable sort by position gives -- merged, and written once, in place.
ISAAC_HD bool leanBuildCandidates(const DevParams &P, const u8 *clusterBcl, const Match *matches, u32 nMatches, bool trim, ClusterFragments &out, const LeanKeyArea &k)
{
    out.nCands[0] = out.nCands[1] = 0; out.cigarUsed = 0; out.flags = 0; out.repeatSeedsCount = 0; out.built = 0;
    out.cands[1] = out.cands[0]; out.candCap[1] = out.candCap[0];
    out.endCyclesMasked[0] = trim ? trimLowQualityEnd(clusterBcl + P.readOffset[0], P.readLength[0], P.baseQualityCutoff) : 0;
    out.endCyclesMasked[1] = (trim && 1 < P.nReads) ? trimLowQualityEnd(clusterBcl + P.readOffset[1], P.readLength[1], P.baseQualityCutoff) : 0;
    if (!nMatches) return false;
    // seedMatchCounts_ / repeatSeedsCount_ (FragmentBuilder.cpp:99-126): a byte per seed in two registers
    u64 countsLow = 0, countsHigh = 0; u32 tooMany = 0;
    bool any = false;
    for (u32 i = 0; i < nMatches; ++i)
    {
        const u64 location = matches[i].location, seedId = matches[i].seedId;
        k.a[i * k.stride] = location; k.b[i * k.stride] = seedId & 0x1ff;       // seed index and strand: all that is read of a seed id
        if (refposIsNoMatch(location)) continue;
        any = true;
        const u32 s = seedIdSeed(seedId);
        if (refposIsTooMany(location)) tooMany |= 1u << s;
        else if (s & 8) countsHigh += u64(1) << (8 * (s & 7)); else countsLow += u64(1) << (8 * s);
    }
    if (!any) return false;
    u32 skipSeeds = tooMany;                                                      // seeds whose matches make no candidates
    u32 repeatSeedsCount = 0;
    for (u32 s = 0; s < P.nSeeds; ++s)
    {
        const u32 count = u32(((s & 8) ? countsHigh : countsLow) >> (8 * (s & 7))) & 0xffu;
        if (count >= P.repeatThreshold) skipSeeds |= 1u << s;
        if ((skipSeeds >> s) & 1) ++repeatSeedsCount;
    }
    out.repeatSeedsCount = repeatSeedsCount;
    // every match becomes its candidate's keys in place; which read it belongs to and whether it counts: two bit masks
    u32 valid = 0, ofRead1 = 0;
    for (u32 i = 0; i < nMatches; ++i)
    {
        const u64 location = k.a[i * k.stride]; const u32 tie = u32(k.b[i * k.stride]);
        if (refposIsNoMatch(location) || refposIsTooMany(location)) continue;
        const u32 s = tie >> 1;
        if ((skipSeeds >> s) & 1) continue;
        const DevSeed &seed = P.seeds[s];
        const u32 r = seed.readIndex;
        if (r >= P.nReads) continue;
        const bool reverse = tie & 1;
        const i64 seedPosition = i64(refposPosition(location));
        const i64 position = reverse ? seedPosition + seed.length + seed.offset - i64(P.readLength[r]) : seedPosition - seed.offset;
        const bool nonUnique = seed.length != 64 && (location & 1);
        k.a[i * k.stride] = leanPositionKey(refposContig(location), position);
        k.b[i * k.stride] = (u64(reverse) << 45) | (u64(seedPosition) << 5) | (u64(s) << 1) | u64(nonUnique);
        valid |= 1u << i; if (r) ofRead1 |= 1u << i;
    }
    bool built = false;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        if (r) { out.cands[1] = out.cands[0] + out.nCands[0]; out.candCap[1] = out.candCap[0] - out.nCands[0]; }
        u32 mine = valid & (r ? ofRead1 : ~ofRead1);
        if (!mine) continue;
        built = true;
        Nibbles order; order.v = 0; u32 n = 0;
        while (mine) { const u32 i = u32(__builtin_ctz(mine)); mine &= mine - 1; order.set(n++, i); }
        leanSortOrder(order, n, k);
        const u32 stored = leanEmitCandidates(P, out.list(r), r, repeatSeedsCount, n, k, NibbleIndex{order});
        out.setListLength(r, stored);
    }
    out.built = built;
    return built;
}

// The same for a cluster with any number of matches that its key area holds (the general form of k_build_fragments): the two sorts are the
// instruction-exact std::sort of sort.h over index arrays -- the match order, then per read the order of the candidates under
// FragmentMetadata::operator< alone, in which equal candidates keep whatever place std::sort gives them -- on keys in the area instead of
// on match and candidate records; candidates are written once, in their final order.  matchOrder, candOrder: nMatches bytes each.
// Precondition: the cluster's slots hold nMatches candidates and a read has at most CAND_CAP of them.
struct MatchKeyLess
{
    LeanKeyArea k;              // a = location, b = seed << 1 | strand
    ISAAC_HD bool operator()(u8 x, u8 y) const { return leanKeyLess(k, x, y); }
};
struct CandKeyLess
{
    LeanKeyArea k;              // a = leanPositionKey, b bit 45 = strand
    ISAAC_HD bool operator()(u8 x, u8 y) const
    {
        const u64 ax = k.a[x * k.stride], ay = k.a[y * k.stride];
        if (ax != ay) return ax < ay;
        return ((k.b[x * k.stride] >> 45) & 1) < ((k.b[y * k.stride] >> 45) & 1);
    }
};
ISAAC_HD bool keyedBuildCandidates(const DevParams &P, const u8 *clusterBcl, const Match *matches, u32 nMatches, bool trim, ClusterFragments &out, const LeanKeyArea &k, u8 *matchOrder, u8 *candOrder)
{
    out.nCands[0] = out.nCands[1] = 0; out.cigarUsed = 0; out.flags = 0; out.repeatSeedsCount = 0; out.built = 0;
    out.cands[1] = out.cands[0]; out.candCap[1] = out.candCap[0];
    out.endCyclesMasked[0] = trim ? trimLowQualityEnd(clusterBcl + P.readOffset[0], P.readLength[0], P.baseQualityCutoff) : 0;
    out.endCyclesMasked[1] = (trim && 1 < P.nReads) ? trimLowQualityEnd(clusterBcl + P.readOffset[1], P.readLength[1], P.baseQualityCutoff) : 0;
    if (!nMatches) return false;
    u64 countsLow = 0, countsHigh = 0; u32 tooMany = 0;
    bool any = false;
    for (u32 i = 0; i < nMatches; ++i)
    {
        const u64 location = matches[i].location, seedId = matches[i].seedId;
        k.a[i * k.stride] = location; k.b[i * k.stride] = seedId & 0x1ff;
        matchOrder[i] = u8(i);
        if (refposIsNoMatch(location)) continue;
        any = true;
        const u32 s = seedIdSeed(seedId);
        if (refposIsTooMany(location)) tooMany |= 1u << s;
        else if (s & 8) countsHigh += u64(1) << (8 * (s & 7)); else countsLow += u64(1) << (8 * s);
    }
    if (!any) return false;
    u32 skipSeeds = tooMany;
    u32 repeatSeedsCount = 0;
    for (u32 s = 0; s < P.nSeeds; ++s)
    {
        const u32 count = u32(((s & 8) ? countsHigh : countsLow) >> (8 * (s & 7))) & 0xffu;
        if (count >= P.repeatThreshold) skipSeeds |= 1u << s;
        if ((skipSeeds >> s) & 1) ++repeatSeedsCount;
    }
    out.repeatSeedsCount = repeatSeedsCount;
    // the reference adds candidates in sorted match order: that order is the input order of the sort by position
    { MatchKeyLess ml; ml.k = k; exactSort(matchOrder, i32(nMatches), ml); }
    // every match becomes its candidate's keys in place; bits 62 / 63 of b: it counts / it belongs to read 1
    for (u32 i = 0; i < nMatches; ++i)
    {
        const u64 location = k.a[i * k.stride]; const u32 tie = u32(k.b[i * k.stride]);
        k.b[i * k.stride] = 0;
        if (refposIsNoMatch(location) || refposIsTooMany(location)) continue;
        const u32 s = tie >> 1;
        if ((skipSeeds >> s) & 1) continue;
        const DevSeed &seed = P.seeds[s];
        const u32 r = seed.readIndex;
        if (r >= P.nReads) continue;
        const bool reverse = tie & 1;
        const i64 seedPosition = i64(refposPosition(location));
        const i64 position = reverse ? seedPosition + seed.length + seed.offset - i64(P.readLength[r]) : seedPosition - seed.offset;
        const bool nonUnique = seed.length != 64 && (location & 1);
        k.a[i * k.stride] = leanPositionKey(refposContig(location), position);
        k.b[i * k.stride] = (u64(1) << 62) | (u64(r) << 63) | (u64(reverse) << 45) | (u64(seedPosition) << 5) | (u64(s) << 1) | u64(nonUnique);
    }
    bool built = false;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        if (r) { out.cands[1] = out.cands[0] + out.nCands[0]; out.candCap[1] = out.candCap[0] - out.nCands[0]; }
        u32 n = 0;
        for (u32 at = 0; at < nMatches; ++at)
        {
            const u32 i = matchOrder[at];
            const u64 b = k.b[i * k.stride];
            if (((b >> 62) & 1) && u32(b >> 63) == r) candOrder[n++] = u8(i);
        }
        if (!n) continue;
        built = true;
        { CandKeyLess cl; cl.k = k; exactSort(candOrder, i32(n), cl); }
        ByteIndex index; index.order = candOrder;
        out.setListLength(r, leanEmitCandidates(P, out.list(r), r, repeatSeedsCount, n, k, index));
    }
    out.built = built;
    return built;
}

// ------------------------------------------------------------------------------------------------------------------
// k_finish_candidates: aligner.h: finishCandidates with the single-indel stage deferred, both lists up to LEAN_LIST_MAX long
ISAAC_HD void leanFinishCandidates(const DevParams &P, ClusterFragments &out, const LeanKeyArea &k)
{
    if (!out.built) return;
    const u32 cigarUsed = 3 * (out.nCands[0] + out.nCands[1]);
    u32 slotBase = 0;                     // the list's first candidate slot: its ungapped CIGAR is at 3 x slot
    for (u32 r = 0; r < P.nReads; ++r)
    {
        const u32 n0 = out.listLength(r);
        if (!n0) continue;
        Cand *store = out.list(r);
        u32 n = leanConsolidate(store, n0, true, k);
        if (P.semialignedGapLimit && n >= 2)
        {
            // sortForSimpleIndels + hasSimpleIndelPair (SimpleIndelAligner.cpp:443-449,460-518): by (contig, unclipped position).  A second
            // consolidation afterwards returns a list without a candidate pair to the order it has now (its members are all different).
            bool sorted = true; u32 reverseMask = 0;
            for (u32 i = 0; i < n; ++i)
            {
                const Cand &c = store[i];
                const i64 unclipped = c.position - candBeginClipped(c, out.cigarPool);
                k.a[i * k.stride] = leanPositionKey(c.contigId, unclipped); k.b[i * k.stride] = 0;
                if (c.reverse) reverseMask |= 1u << i;
                if (i && k.a[i * k.stride] < k.a[(i - 1) * k.stride]) sorted = false;
            }
            Nibbles order; order.v = NIBBLES_IDENTITY;
            if (!sorted) leanSortOrder(order, n, k);
            bool pair = false;
            for (u32 t = 1; t < n && !pair; ++t)
            {
                const u32 ih = order.get(t - 1), it = order.get(t);
                const u64 ha = k.a[ih * k.stride], ta = k.a[it * k.stride];
                if ((ha >> 41) != (ta >> 41) || (((reverseMask >> ih) ^ (reverseMask >> it)) & 1)) continue;
                const i64 distance = i64(ta) - i64(ha);          // same contig: the difference of the position parts
                if (!distance) continue;
                pair = (distance < 0 ? -distance : distance) < i64(P.semialignedGapLimit);
            }
            if (pair)
            {
                out.flags |= CLUSTER_INDEL_PENDING << r;            // the list is left in this order for finishSimpleIndels
                if (!sorted) leanApplyOrder(store, order, n);
            }
        }
        out.setListLength(r, n);
        slotBase += n0;
    }
    (void)slotBase;
    out.cigarUsed = cigarUsed;
}

// ------------------------------------------------------------------------------------------------------------------
// k_finish_fragments: aligner.h: finishFragments on the results of the flat banded Smith-Waterman pass (in countGappedJobs order),
// both lists up to LEAN_LIST_MAX long
ISAAC_HD void leanFinishFragments(const DevParams &P, ClusterFragments &out, const GappedResult *results, const LeanKeyArea &k, u32 &bswJobs, u32 &bswAccepted, u32 &candidates)
{
    if (!out.built) return;
    CigarPool pool; pool.words = out.cigarPool; pool.used = out.cigarUsed; pool.capacity = out.cigarCap; pool.overflow = 0;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        const u32 n = out.listLength(r);
        Cand *store = out.list(r);
        bool changed = false;
        for (u32 i = 0; results && i < n; ++i)
        {
            Cand &f = store[i];
            const u32 mismatchCount = f.mismatchCount;
            if (!(BSW_MISMATCHES_CUTOFF < mismatchCount)) continue;
            const GappedResult &g = *results++;
            ++bswJobs;
            if (0xffffffffu == g.nCigar) { out.flags |= CLUSTER_OVERFLOW; continue; }
            const u32 matchCount = g.matchCount;
            if (matchCount && matchCount + BSW_WIDEST_GAP_SIZE > candObservedLength(f) && (g.out.mismatchCount <= P.gappedMismatchesMax) &&
                (mismatchCount > g.out.mismatchCount) && lpLess(f.logProbability, g.out.logProbability))
            {
                f = g.out;
                f.cigarOffset = pool.used;
                for (u32 w = 0; w < g.nCigar; ++w) pool.push(g.cigar[w]);
                ++bswAccepted;
                changed = true;
            }
        }
        (void)changed;
        const u32 m = leanConsolidate(store, n, true, k);          // (a list without an accepted gapped alignment is consolidated already: one pass over its keys)
        out.setListLength(r, m);
        candidates += m;
    }
    out.cigarUsed = pool.used;
    if (pool.overflow) out.flags |= CLUSTER_OVERFLOW;
}

} // namespace isaac
