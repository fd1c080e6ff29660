// The probability sums of TemplateBuilder (sumUniqueShadowProbabilities / sumUniquePairProbabilities, TemplateBuilder.cpp:694-714,
// and the running sum of rescueShadow, :552-560) and the outcome of every ShadowAligner::rescueShadow call of a cluster
// (ShadowAligner.cpp:232-291), computed by a group of lanes per cluster from what the flat rescue kernels left in HBM.
// k_select then runs the template logic on numbers (RESCUE_PRECOMPUTED) instead of building, copying and sorting lists per thread.
//
// A list is never materialised: element r of the shadow list of rescue problem j is the aligned candidate of rank r (its gapped
// retry where ShadowAligner accepted one), so the sort keys are gathered straight into the group's LDS arrays, ordered by a
// bitonic network over 16-bit indices, and summed in sorted order by the reference's rule (the first of every run of equal
// elements).  The order used is the total order that refines the reference's epsilon comparators; a list with a "near tie"
// (same position, probabilities different but within 1e-7), where the two can disagree, sends the cluster to the
// wave-per-cluster pass, which has the instruction-exact std::sort replica.  So do clusters whose lists exceed the group's
// capacity and clusters for which a capacity of the flat pass was exceeded.
#pragma once
#include "template.h"

namespace isaac
{

// -DISAAC_PROFILE_SUMS: the first clusters of the HBM tier print where their time goes (100 MHz ticks), timing experiments only
#if defined(ISAAC_PROFILE_SUMS) && defined(__HIP_DEVICE_COMPILE__)
#define SUMS_T0() long long sums_t = wall_clock64()
#define SUMS_T(name, n) do { if (g.radix.counts && 0 == g.lane && blockIdx.x < 6) { const long long t = wall_clock64(); printf("sums block %u %s n %u: %lld\n", blockIdx.x, name, u32(n), t - sums_t); sums_t = wall_clock64(); } } while (0)
#else
#define SUMS_T0()
#define SUMS_T(name, n)
#endif

// `cap` entries each, in LDS on the device
struct SumKeys { u64 *pos1, *pos2; double *lp, *term; u32 *obs1, *obs2; u16 *idx; u32 cap; };
ISAAC_HD u64 sumKeysBytes(u32 cap) { return u64(cap) * (8 + 8 + 8 + 8 + 4 + 4 + 2); }
ISAAC_HD void sumKeysBind(SumKeys &k, void *base, u32 cap)
{
    u8 *p = static_cast<u8 *>(base);
    k.cap = cap;
    k.pos1 = reinterpret_cast<u64 *>(p); p += u64(cap) * 8; k.pos2 = reinterpret_cast<u64 *>(p); p += u64(cap) * 8;
    k.lp = reinterpret_cast<double *>(p); p += u64(cap) * 8; k.term = reinterpret_cast<double *>(p); p += u64(cap) * 8;
    k.obs1 = reinterpret_cast<u32 *>(p); p += u64(cap) * 4; k.obs2 = reinterpret_cast<u32 *>(p); p += u64(cap) * 4;
    k.idx = reinterpret_cast<u16 *>(p);
}

// the lanes working on one cluster: one wavefront, or a whole workgroup (block = true)
// radix: work area of the radix ordering used for long lists (counts: 16 x lanes, totals: lanes, vary: 2, alt: as many entries as the
// key arrays), or all NULL
struct SumRadix { u16 *counts; u32 *totals; u64 *vary; u16 *alt; u8 *digits; u32 digitsCap;      // digits (optional, one byte per entry, close memory): see radixOrder
                  u16 *closeIdx = nullptr; u32 closeIdxCap = 0; };      // optional: two index arrays of closeIdxCap entries in close memory, used instead of k.idx / alt for lists that fit
// sumTile: LDS room for sumTileCap terms when the key arrays are not in LDS themselves (the final additions are a chain of
// dependent loads otherwise), or NULL
struct SumGroup { u32 lanes, lane; bool block; SumRadix radix; u32 radixMin; double *sumTile; u32 sumTileCap; };
ISAAC_HD void groupSync(const SumGroup &g)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (g.block) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#endif
    (void)g;
}
ISAAC_HD bool groupAny(const SumGroup &g, bool v, u32 *scratch)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (!g.block)
    {
        if (g.lanes >= 64) return __ballot(v) != 0;
        const u32 groupBase = (threadIdx.x & 63u) & ~(g.lanes - 1);            // a part of a wavefront: its lanes' votes only
        return 0 != ((__ballot(v) >> groupBase) & ((1ull << g.lanes) - 1));
    }
    if (0 == g.lane) *scratch = 0;
    __syncthreads();
    if (v) *scratch = 1;
    __syncthreads();
    const bool r = 0 != *scratch;
    __syncthreads();
    return r;
#else
    (void)g; (void)scratch; return v;
#endif
}

// total orders refining shadowProbLess / pairProbLess (keyLess of template.h), on entries of the key arrays
ISAAC_HD bool sumKeyLess(const SumKeys &k, bool pairs, u32 a, u32 b)
{
    if (k.pos1[a] != k.pos1[b]) return k.pos1[a] < k.pos1[b];
    if (k.pos2[a] != k.pos2[b]) return k.pos2[a] < k.pos2[b];
    if (k.lp[a] != k.lp[b]) return pairs ? k.lp[b] < k.lp[a] : k.lp[a] < k.lp[b];      // pairs: higher probability first
    if (k.obs1[a] != k.obs1[b]) return k.obs1[a] < k.obs1[b];
    if (k.obs2[a] != k.obs2[b]) return k.obs2[a] < k.obs2[b];
    return a < b;
}
ISAAC_HD bool sumKeyNearTie(const SumKeys &k, u32 a, u32 b) { return k.pos1[a] == k.pos1[b] && k.pos2[a] == k.pos2[b] && k.lp[a] != k.lp[b] && lpEquals(k.lp[a], k.lp[b]); }
ISAAC_HD bool sumKeyEqual(const SumKeys &k, u32 a, u32 b)
{ return k.pos1[a] == k.pos1[b] && k.pos2[a] == k.pos2[b] && lpEquals(k.lp[a], k.lp[b]) && k.obs1[a] == k.obs1[b] && k.obs2[a] == k.obs2[b]; }

// what follows the positions in the total order
ISAAC_HD bool sumKeyRestLess(const SumKeys &k, bool pairs, u32 a, u32 b)
{
    if (k.lp[a] != k.lp[b]) return pairs ? k.lp[b] < k.lp[a] : k.lp[a] < k.lp[b];
    if (k.obs1[a] != k.obs1[b]) return k.obs1[a] < k.obs1[b];
    if (k.obs2[a] != k.obs2[b]) return k.obs2[a] < k.obs2[b];
    return a < b;
}

// The same order as the bitonic network of uniqueSortedSum for lists of thousands of entries: stable least-significant-digit radix
// passes (4 bits, every lane a contiguous slice of the list, digit-major counts) over the two positions -- only over the nibbles that
// differ inside the list, typically six to ten of the 32 -- then every entry finds its place inside its run of equal positions by
// counting.  A few dozen passes with independent, mostly coalesced loads instead of 120 dependent exchange steps.
// Returns the array that holds the order (k.idx, g.radix.alt or one of the close index arrays).
ISAAC_HD const u16 *radixOrder(SumKeys &k, u32 n, bool pairs, const SumGroup &g)
{
    SUMS_T0();
    SumRadix r = g.radix;
    if (n > r.digitsCap) r.digits = nullptr;          // the close array holds digitsCap entries: longer lists gather from the key arrays
    const u32 per = (n + g.lanes - 1) / g.lanes, begin = imin(n, g.lane * per), end = imin(n, begin + per);
    u16 *src = k.idx, *dst = r.alt;
    // the two index arrays of the passes in close memory when the list fits: a pass then reads and writes them there instead of going to the L2 for every
    // entry, twice (5 - 6 us a pass instead of 11 for the 8 000-entry lists that are the usual case of this tier)
    if (r.closeIdx && n <= r.closeIdxCap) { src = r.closeIdx; dst = r.closeIdx + r.closeIdxCap; }
    for (u32 i = begin; i < end; ++i) src[i] = u16(i);
    if (0 == g.lane) { r.vary[0] = 0; r.vary[1] = 0; }
    groupSync(g);
    {   // bits that differ somewhere in the list
        u64 v1 = 0, v2 = 0; const u64 f1 = k.pos1[0], f2 = k.pos2[0];
        for (u32 i = begin; i < end; ++i) { v1 |= k.pos1[i] ^ f1; v2 |= k.pos2[i] ^ f2; }
#if defined(__HIP_DEVICE_COMPILE__)
        if (v1) atomicOr(reinterpret_cast<unsigned long long *>(&r.vary[0]), (unsigned long long)v1);
        if (v2) atomicOr(reinterpret_cast<unsigned long long *>(&r.vary[1]), (unsigned long long)v2);
#else
        r.vary[0] |= v1; r.vary[1] |= v2;
#endif
    }
    groupSync(g);
    for (u32 word = 0; word < 2; ++word)
    {
        const u64 *key = word ? k.pos1 : k.pos2;         // least significant key first
        const u64 vary = r.vary[word ? 0 : 1];
        for (u32 shift = 0; shift < 64; shift += 4)
        {
            if (!((vary >> shift) & 15)) continue;
            for (u32 d = 0; d < 16; ++d) r.counts[d * g.lanes + g.lane] = 0;
            // The digit of every entry, in entry order: one coalesced pass over the key word.  The two passes below then look the digit of
            // src[i] up in close memory instead of gathering key[src[i]] from the (HBM-resident) key array, twice.
            if (r.digits) { for (u32 i = g.lane; i < n; i += g.lanes) r.digits[i] = u8((key[i] >> shift) & 15); groupSync(g); }
            for (u32 i = begin; i < end; ++i) ++r.counts[(r.digits ? u32(r.digits[src[i]]) : u32((key[src[i]] >> shift) & 15)) * g.lanes + g.lane];
            groupSync(g);
            {   // exclusive prefix over the 16 x lanes counts, digit-major: a lane sums 16 consecutive ones, the lanes' sums are scanned
                u32 sum = 0;
                for (u32 e = 0; e < 16; ++e) sum += r.counts[g.lane * 16 + e];
                u32 running;
#if defined(__HIP_DEVICE_COMPILE__)
                if (g.block && 0 == (g.lanes & 63u))
                {   // inside the wavefronts by shuffles, then the wavefronts' totals (at most 16 of them, read by everyone): two barriers
                    // instead of the twenty of a scan in steps over 1024 lanes -- a list of thousands of entries has some ten such passes,
                    // three lists a cluster, and nothing else to do while it waits
                    u32 incl = sum;
                    for (u32 o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((g.lane & 63u) >= o) incl += v; }
                    if (63u == (g.lane & 63u)) r.totals[g.lane >> 6] = incl;
                    groupSync(g);
                    u32 before = 0;
                    for (u32 w = 0; w < (g.lane >> 6); ++w) before += r.totals[w];
                    groupSync(g);
                    running = before + incl - sum;
                }
                else
#endif
                {
                    r.totals[g.lane] = sum;
                    groupSync(g);
                    for (u32 step = 1; step < g.lanes; step <<= 1)
                    {
                        const u32 add = g.lane >= step ? r.totals[g.lane - step] : 0;
                        groupSync(g);
                        r.totals[g.lane] += add;
                        groupSync(g);
                    }
                    running = r.totals[g.lane] - sum;
                }
                for (u32 e = 0; e < 16; ++e) { const u32 c = r.counts[g.lane * 16 + e]; r.counts[g.lane * 16 + e] = u16(running); running += c; }
            }
            groupSync(g);
            for (u32 i = begin; i < end; ++i) { const u16 e = src[i]; dst[r.counts[(r.digits ? u32(r.digits[e]) : u32((key[e] >> shift) & 15)) * g.lanes + g.lane]++] = e; }
            groupSync(g);
            u16 *t = src; src = dst; dst = t;
        }
    }
    SUMS_T("radix passes", n);
    // inside a run of equal positions: the place of an entry is the number of the run's entries before it.  With the digit array at hand
    // it first records which entries share their positions with their predecessor (independent loads, a lane per entry): the entries that
    // are runs of their own -- nearly all -- are in place already and need none of the dependent look-ups below
    u8 *same = r.digits;
    if (same)
    {
        for (u32 i = g.lane; i < n; i += g.lanes)
        {
            const u32 e = src[i], p = i ? src[i - 1] : e;
            same[i] = (i && k.pos1[e] == k.pos1[p] && k.pos2[e] == k.pos2[p]) ? 1 : 0;
        }
        groupSync(g);
    }
    for (u32 i = begin; i < end; ++i)
    {
        const u32 e = src[i];
        if (same && !same[i] && !(i + 1 < n && same[i + 1])) { dst[i] = u16(e); continue; }
        const u64 p1 = k.pos1[e], p2 = k.pos2[e];
        u32 lo = i; while (lo && k.pos1[src[lo - 1]] == p1 && k.pos2[src[lo - 1]] == p2) --lo;
        u32 before = 0;
        for (u32 j = lo; j < n; ++j)
        {
            const u32 o = src[j];
            if (k.pos1[o] != p1 || k.pos2[o] != p2) break;
            before += sumKeyRestLess(k, pairs, o, e) ? 1u : 0u;
        }
        dst[lo + before] = u16(e);
    }
    groupSync(g);
    SUMS_T("runs", n);
    return dst;
}

// Sum of exp(lp) over the first element of every run of equal keys of entries [0, n), in sorted order.  false: a near tie.
ISAAC_HD bool uniqueSortedSum(SumKeys &k, u32 n, bool pairs, const SumGroup &g, u32 *scratch, double &sum)
{
    sum = 0.0;
    if (!n) return true;
    SUMS_T0();
    const u16 *order = k.idx;
    const u8 *same = nullptr;                 // radixOrder leaves "same positions as the predecessor in the order" per entry when it has room for it
    if (g.radix.counts && n >= g.radixMin) { order = radixOrder(k, n, pairs, g); same = n <= g.radix.digitsCap ? g.radix.digits : nullptr; }
    else
#if defined(__HIP_DEVICE_COMPILE__)
    if (n <= g.lanes && !g.sumTile)      // (a wavefront or a workgroup; not the HBM tier, whose keys are not in LDS)
    {   // one entry per lane: its place in the order is the number of entries before it, counted against every entry in turn (the
        // LDS reads of entry j are the same address for all lanes); a fraction of the instructions of the network below
        const u32 i = g.lane < n ? g.lane : 0;
        const u64 p1 = k.pos1[i], p2 = k.pos2[i]; const double lp = k.lp[i]; const u32 o1 = k.obs1[i], o2 = k.obs2[i];
        u32 rank = 0;
        if (pairs)
            for (u32 j = 0; j < n; ++j)
            {
                const u64 q1 = k.pos1[j], q2 = k.pos2[j]; const double ql = k.lp[j]; const u32 r1 = k.obs1[j], r2 = k.obs2[j];
                const bool tail = r1 < o1 || (r1 == o1 && (r2 < o2 || (r2 == o2 && j < i)));
                const bool mid = q2 < p2 || (q2 == p2 && (lp < ql || (ql == lp && tail)));
                rank += (q1 < p1 || (q1 == p1 && mid)) ? 1u : 0u;
            }
        else
            for (u32 j = 0; j < n; ++j)
            {   // shadow lists: the second position and the second length are 0 in every entry (sumKeyFromCand)
                const u64 q1 = k.pos1[j]; const double ql = k.lp[j]; const u32 r1 = k.obs1[j];
                const bool tail = r1 < o1 || (r1 == o1 && j < i);
                rank += (q1 < p1 || (q1 == p1 && (ql < lp || (ql == lp && tail)))) ? 1u : 0u;
            }
        if (g.lane < n) k.idx[rank] = u16(i);
        groupSync(g);
    }
    else
#endif
    {
    u32 m = 1; while (m < n) m <<= 1;
    for (u32 i = g.lane; i < m; i += g.lanes) k.idx[i] = i < n ? u16(i) : u16(0xffff);
    groupSync(g);
    for (u32 kk = 2; kk <= m; kk <<= 1)
        for (u32 j = kk >> 1; j > 0; j >>= 1)
        {
            for (u32 t = g.lane; t < (m >> 1); t += g.lanes)
            {
                const u32 i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const u16 a = k.idx[i], b = k.idx[i + j];
                // 0xffff pads the network: greater than every element
                const bool aLessB = (a != 0xffff) && (b == 0xffff || sumKeyLess(k, pairs, a, b));
                const bool bLessA = (b != 0xffff) && (a == 0xffff || sumKeyLess(k, pairs, b, a));
                if ((0 == (i & kk)) ? bLessA : aLessB) { k.idx[i] = b; k.idx[i + j] = a; }
            }
            groupSync(g);
        }
    }
    SUMS_T("order", n);
    bool nearTie = false;
    for (u32 i = g.lane; i < n; i += g.lanes)
    {
        const u32 cur = order[i];
        bool dup = false;
        if (i && (!same || same[i])) { const u32 prev = order[i - 1]; nearTie |= sumKeyNearTie(k, prev, cur); dup = sumKeyEqual(k, prev, cur); }   // both need equal positions
        k.term[i] = dup ? 0.0 : exp(k.lp[cur]);       // without near ties "equal to the first of the run" is "equal to the predecessor"; x + 0.0 == x
    }
    if (groupAny(g, nearTie, scratch)) return false;
    groupSync(g);
    SUMS_T("terms", n);
    if (g.sumTile)
    {   // the terms pass through LDS a tile at a time; the first wavefront adds them up, the others wait
        double acc = 0.0;
        for (u32 base = 0; base < n; base += g.sumTileCap)
        {
            const u32 m = imin(g.sumTileCap, n - base);
            for (u32 i = g.lane; i < m; i += g.lanes) g.sumTile[i] = k.term[base + i];
            groupSync(g);
            if (g.lane < 64) acc = addInOrder(acc, g.sumTile, m);
            groupSync(g);
        }
        if (0 == g.lane) g.sumTile[0] = acc;
        groupSync(g);
        sum = g.sumTile[0];
        groupSync(g);
        SUMS_T("add", n);
        return true;
    }
    sum = addInOrder(sum, k.term, n);                 // the additions in sequence: their order is part of the result (every lane: same value)
    groupSync(g);
    return true;
}

ISAAC_HD void sumKeyFromCand(SumKeys &k, u32 at, const Cand &c)
{ const ShadowProb p = makeShadowProb(c); k.pos1[at] = p.pos; k.pos2[at] = 0; k.lp[at] = p.logProbability; k.obs1[at] = u32(p.observedLength); k.obs2[at] = 0; }

enum { SUMS_DONE = 0, SUMS_TOO_LARGE = 1, SUMS_RESIDUAL = 2 /* a capacity of the flat pass was exceeded */, SUMS_NEAR_TIE = 3 };

// what the flat rescue kernels left for one cluster
struct SumInputs { RescueJob *jobs; u32 nJobs; const Cand *shadowCands; const u32 *candRank; const GappedResult *gappedResults; GappedJob *gappedJobs; const u32 *shadowCigars /* 3 words per candidate slot */; };

// the accept rule of ShadowAligner.cpp:243-262 for the gapped retry `g` of the shadow `fragment`
ISAAC_HD bool gappedRetryAccepted(const DevParams &P, const Cand &fragment, const GappedResult &g)
{
    const Cand &tmp = g.out;
    return 0xffffffffu != g.nCigar && g.matchCount && g.matchCount + BSW_WIDEST_GAP_SIZE > candObservedLength(fragment) && (tmp.mismatchCount <= P.gappedMismatchesMax) &&
           (fragment.mismatchCount > tmp.mismatchCount) && lpLess(fragment.logProbability, tmp.logProbability);
}

// second half of ShadowAligner::rescueShadow (ShadowAligner.cpp:232-291) for one problem, on numbers only: how long the list is,
// whether the call succeeds, which retries are accepted (GappedJob::accepted = 1) and which shadow ends up in front.  false: the
// wave-per-cluster pass has to do it (a capacity of the flat pass was exceeded).
ISAAC_HD bool finishRescueFlat(const DevParams &P, RescueJob &job, const SumInputs &in, u32 &retries)
{
    // What the problem's record gains is put together here and stored once, at the end: written field by field as it became known, the record's two lines
    // went to device memory more than once (the walk over the retries below is a chain of dependent loads, longer than a line stays in the L2 of a busy kernel).
    struct Outcome
    {
        u32 take = 0, rescued = 0, finalBestRank = 0, finalBestSlot = 0, finalBestGapped = 0xffffffffu; RescueOutcome out;
        ISAAC_HD void store(RescueJob &j) const { j.take = take; j.rescued = rescued; j.finalBestRank = finalBestRank; j.finalBestSlot = finalBestSlot; j.finalBestGapped = finalBestGapped; j.out = out; }
    } o;
    o.out.logProbability = 0.0; o.out.smithWatermanScore = 0; o.out.editDistance = 0; o.out.mapped = 0; o.out.mismatchCount = 0; o.out.matchesInARow = 0; o.out.rescued = 0;
    o.out.pad[0] = o.out.pad[1] = o.out.pad[2] = 0;
    const u32 valid = job.valid, fallback = job.fallback, nGappedJobs = job.nGapped, gappedBase = job.gappedBase, nAligned = job.nAligned, lastAligned = job.lastAligned, nCands = job.nCands;
    if (!valid) { o.store(job); return true; }
    if (fallback || (nGappedJobs && 0xffffffffu == gappedBase)) { o.store(job); return false; }
    const bool full = nAligned - lastAligned >= SHADOW_LIST_MAX && nCands != 0;   // the reference gives up when the list is full and another candidate aligns
    o.take = imin(nAligned, SHADOW_LIST_MAX);
    if (full || !nAligned) { o.store(job); return true; }
    u32 best = job.bestRank, bestSlot = job.bestSlot, bestGapped = 0xffffffffu;
    double bestLp = in.shadowCands[bestSlot].logProbability;
    for (u32 kk = 0; kk < nGappedJobs; ++kk)
    {
        const GappedResult &g = in.gappedResults[gappedBase + kk];
        GappedJob &gj = in.gappedJobs[gappedBase + kk];
        const u32 slot = gj.tag, i = in.candRank[slot];
        const Cand &fragment = in.shadowCands[slot];
        ++retries;
        gj.accepted = 0;
        if (0xffffffffu == g.nCigar) { o.store(job); return false; }        // CIGAR longer than the result record holds
        const Cand &tmp = g.out;
        if (gappedRetryAccepted(P, fragment, g))
        {
            gj.accepted = 1;
            if (i == best) { bestLp = tmp.logProbability; bestGapped = gappedBase + kk; bestSlot = slot; }
            else if (lpLess(bestLp, tmp.logProbability)) { best = i; bestLp = tmp.logProbability; bestGapped = gappedBase + kk; bestSlot = slot; }
        }
    }
    o.rescued = 1; o.finalBestRank = best; o.finalBestSlot = bestSlot; o.finalBestGapped = bestGapped;
    {   // the best shadow as the template stage will ask about it (template_lean.h: leanConsiderRescued)
        const bool gapped = 0xffffffffu != bestGapped;
        const Cand &b = gapped ? in.gappedResults[bestGapped].out : in.shadowCands[bestSlot];
        const u32 *cigar = gapped ? in.gappedResults[bestGapped].cigar : in.shadowCigars + u64(bestSlot) * 3;
        const u32 n = gapped ? (in.gappedResults[bestGapped].nCigar & 0xffffu) : u32(b.cigarLength);
        u32 mapped = 0;
        for (u32 i = 0; i < n; ++i) if (OP_ALIGN == cigarCode(cigar[i])) mapped += cigarLen(cigar[i]);
        o.out.logProbability = b.logProbability; o.out.smithWatermanScore = b.smithWatermanScore; o.out.editDistance = b.editDistance; o.out.mapped = u16(mapped);
        o.out.mismatchCount = b.mismatchCount; o.out.matchesInARow = b.matchesInARow; o.out.rescued = 1;
    }
    o.store(job);
    return true;
}

// the shadows of one problem, f(rank, cand) for every element of its list (any order; all lanes of the group take part)
template <typename F>
ISAAC_HD void forEachShadow(const RescueJob &job, const SumInputs &in, const SumGroup &g, F f)
{
    for (u32 c = g.lane; c < job.nCands; c += g.lanes)
    {
        const u32 slot = job.candBase + c;
        const Cand &cand = in.shadowCands[slot];
        if (!candAligned(cand)) continue;
        const u32 r = in.candRank[slot];
        if (r < job.take) f(r, cand, false);
    }
    if (!job.rescued) return;
    groupSync(g);                                         // the accepted retries replace what the loop above wrote
    for (u32 kk = g.lane; kk < job.nGapped; kk += g.lanes)
    {
        const GappedJob &gj = in.gappedJobs[job.gappedBase + kk];
        if (!gj.accepted) continue;
        const u32 r = in.candRank[gj.tag];
        if (r < job.take) f(r, in.gappedResults[job.gappedBase + kk].out, true);
    }
}

// Everything k_select needs to know about the mate rescues of one cluster.  `first`: the jobs have not been finished yet (the
// larger group that retries a SUMS_TOO_LARGE cluster skips that step).
// `part`: one of the cluster's three lists (0, 1: the shadow sums of either side; 2: the pair sum or the running sum) or all of them -- the lists are
// independent once the problems are finished, and a cluster of a repeat family is a millisecond of barriers and memory round trips per list: the
// tiers for long lists give every list a workgroup of its own.  The capacity check looks at all three lists whichever part is asked for.
static const u32 SUMS_ALL_PARTS = 3;
ISAAC_HD u32 clusterSums(const DevParams &P, const ClusterFragments &f, const SumInputs &in, SumKeys &k, const SumGroup &g, u32 *scratch, bool first, ClusterSums &out, Counters &cnt,
                         u32 part = SUMS_ALL_PARTS)
{
    out.shadow[0] = out.shadow[1] = out.pair = out.ordered = 0.0;
    if (first)
    {   // every lane does this for itself (the same loads, the same stores of the same values): nothing to hand from lane to lane
        bool ok = true; u32 retries = 0;
        for (u32 j = 0; j < in.nJobs && ok; ++j) ok = finishRescueFlat(P, in.jobs[j], in, retries);
        if (!ok) return SUMS_RESIDUAL;
        if (0 == g.lane) cnt.rescueBsw += retries;
    }
    SUMS_T0();
    const u32 nCands[2] = { f.nCands[0], f.nCands[1] };
    u32 shadows[2] = { 0, 0 };
    for (u32 j = 0; j < in.nJobs; ++j) shadows[(in.jobs[j].shadowReadIndex + 1) % 2] += in.jobs[j].take;
    const bool bothReads = nCands[0] && nCands[1];
    if (shadows[0] + nCands[1] > k.cap || shadows[1] + nCands[0] > k.cap || shadows[0] + shadows[1] > k.cap) return SUMS_TOO_LARGE;
    // sumUniqueShadowProbabilities of either side: the shadows its orphans rescued + the seeded candidates of the other read
    for (u32 side = 0; side < 2; ++side)
    {
        if (SUMS_ALL_PARTS != part && side != part) continue;
        u32 base = 0;
        for (u32 j = 0; j < in.nJobs; ++j)
        {
            const RescueJob &job = in.jobs[j];
            if ((job.shadowReadIndex + 1u) % 2 != side || !job.take) continue;
            forEachShadow(job, in, g, [&](u32 r, const Cand &c, bool) { sumKeyFromCand(k, base + r, c); });
            base += job.take;
        }
        for (u32 i = g.lane; i < nCands[1 - side]; i += g.lanes) sumKeyFromCand(k, base + i, f.list(1 - side)[i]);
        groupSync(g);
        SUMS_T("gather", in.nJobs);
        if (!uniqueSortedSum(k, base + nCands[1 - side], false, g, scratch, out.shadow[side])) return SUMS_NEAR_TIE;
    }
    if (SUMS_ALL_PARTS != part && 2 != part) return SUMS_DONE;
    if (bothReads)
    {   // sumUniquePairProbabilities: every orphan with every shadow it rescued, read 1's alignment first
        u32 base = 0;
        for (u32 j = 0; j < in.nJobs; ++j)
        {
            const RescueJob &job = in.jobs[j];
            if (!job.take) continue;
            const u32 side = (job.shadowReadIndex + 1u) % 2;
            const ShadowProb o = makeShadowProb(f.list(side)[job.orphanListIndex]);
            forEachShadow(job, in, g, [&](u32 r, const Cand &c, bool)
            {
                const ShadowProb s = makeShadowProb(c);
                const ShadowProb &r1 = side ? s : o, &r2 = side ? o : s;
                const u32 at = base + r;
                k.pos1[at] = r1.pos; k.pos2[at] = r2.pos; k.lp[at] = r1.logProbability + r2.logProbability; k.obs1[at] = u32(r1.observedLength); k.obs2[at] = u32(r2.observedLength);
            });
            base += job.take;
        }
        groupSync(g);
        if (!uniqueSortedSum(k, base, true, g, scratch, out.pair)) return SUMS_NEAR_TIE;
    }
    else
    {   // TemplateBuilder::rescueShadow's running sum over the shadow lists in list order: the best shadow of a successful rescue
        // has changed places with the first one
        u32 base = 0;
        for (u32 j = 0; j < in.nJobs; ++j)
        {
            const RescueJob &job = in.jobs[j];
            if (!job.take) continue;
            const u32 side = (job.shadowReadIndex + 1u) % 2;
            const double orphanLp = f.list(side)[job.orphanListIndex].logProbability;
            const u32 best = job.rescued ? job.finalBestRank : 0;
            forEachShadow(job, in, g, [&](u32 r, const Cand &c, bool) { k.term[base + (r == best ? 0 : 0 == r ? best : r)] = exp(orphanLp + c.logProbability); });
            base += job.take;
        }
        groupSync(g);
        double sum = 0.0;
        if (g.sumTile)
        {
            for (u32 b0 = 0; b0 < base; b0 += g.sumTileCap)
            {
                const u32 m = imin(g.sumTileCap, base - b0);
                for (u32 i = g.lane; i < m; i += g.lanes) g.sumTile[i] = k.term[b0 + i];
                groupSync(g);
                if (g.lane < 64) sum = addInOrder(sum, g.sumTile, m);
                groupSync(g);
            }
            if (0 == g.lane) g.sumTile[0] = sum;
            groupSync(g);
            sum = g.sumTile[0];
        }
        else sum = addInOrder(sum, k.term, base);
        out.ordered = sum;
        groupSync(g);
    }
    return SUMS_DONE;
}

} // namespace isaac
