// Gap realignment of the BAM stage (--realign-gaps sample|project|all, with or without --realign-vigorously): build::GapRealigner::realign
// (lib/build/GapRealigner.cpp:1053-1268) for one fragment per thread, against the gaps every fragment of the same contig brought in
// (RealignerGaps, include/build/GapRealigner.hh:37-128; BinSorter::collectGaps, lib/build/BinSorter.cpp:387-403).
//
// Thread-serial ISAAC_HD code like aligner.h / template.h: the kernels of bam_kernels.h call it, tests/hostemu compiles it for the CPU.
// Without --realign-vigorously a fragment with more than MAX_GAPS_AT_A_TIME gaps in reach is left alone (:1116-1120): ten gaps and 2^10 choices.  With it
// (round 6) the fragment is tried against whatever is in reach -- gapRealigner::OverlappingGapsFilter gives up beyond MAX_TRACKED_DELETIONS = 30 gaps
// (OverlappingGapsFilter.hh:36-43), the choices looked at stay 2^10 -- and a fragment that was realigned is tried again until nothing improves (:1241).  Thirty
// gaps and a CIGAR of a few dozen operations bound everything here; the work areas are small private arrays.
#pragma once
#include "types.h"

namespace isaac
{

// gapRealigner::Gap (include/build/gapRealigner/Gap.hh:31-78): pos = ReferencePosition value; length > 0 deletion, < 0 insertion
struct RealignGap { u64 pos; i32 length; u32 pad; };
ISAAC_HD bool rgIsInsertion(const RealignGap &g) { return 0 > g.length; }
ISAAC_HD bool rgIsDeletion(const RealignGap &g) { return 0 < g.length; }
ISAAC_HD u32 rgLength(const RealignGap &g) { return u32(g.length < 0 ? -g.length : g.length); }
ISAAC_HD u64 rpPlus(u64 position, i64 offset) { return position + (u64(offset) << 1); }                 // ReferencePosition::operator+ (ReferencePosition.hh:139-172)
ISAAC_HD i64 rpMinus(u64 l, u64 r) { return i64(refposPosition(l)) - i64(refposPosition(r)); }          // ReferencePosition - ReferencePosition
ISAAC_HD u64 rgEndPos(const RealignGap &g, bool fatInsertions) { return (rgIsDeletion(g) || fatInsertions) ? rpPlus(g.pos, rgLength(g)) : g.pos; }
ISAAC_HD bool rgLess(const RealignGap &l, const RealignGap &r) { return l.pos < r.pos || (l.pos == r.pos && l.length < r.length); }   // orderByGapStartAndTypeLength
// the sort key of a gap: ReferencePosition value, then the signed length
ISAAC_HD u64 rgLengthKey(i32 length) { return u64(u32(length) ^ 0x80000000u); }

// the gaps of a gap group (one sample): sorted by (begin, length), and the deletions again sorted by their end (RealignerGaps::finalizeGaps)
struct RealignerGapsView { const RealignGap *gaps; u32 nGaps; const RealignGap *deletionEnds; u32 nDeletionEnds; };

static const u32 RG_MAX_GAPS_AT_A_TIME = 10;       // GapRealigner::MAX_GAPS_AT_A_TIME
static const u32 RG_TRACKED_GAPS_MAX = 30;         // OverlappingGapsFilter::MAX_TRACKED_DELETIONS: more gaps than this and no choice is looked at
static const u32 RG_FOUND_CAP = 2 * RG_TRACKED_GAPS_MAX;   // starts + ends before the duplicates go: more cannot leave thirty or fewer
static const u32 RG_CIGAR_CAP = 64;

struct RealignParams { u32 mismatchCost, gapOpenCost, gapExtendCost, realignDodgyFragments, clipSemialigned, realignGapsVigorously; };   // BinSorter.hh:96-98: 3, 4, 0

// the fields of io::FragmentAccessor the realigner reads and changes
struct RealignFragment
{
    u64 fStrandPosition, mateFStrandPosition; u32 observedLength, flags; u16 lowClipped, highClipped, alignmentScore, templateAlignmentScore, readLength, editDistance;
    const u8 *bcl;          // the read's BCL bytes as sequenced; basesBegin()[i] of the reference is storedBase(i)
};
ISAAC_HD u32 rfLeftClipped(const RealignFragment &f) { return (f.flags & 8) ? f.highClipped : f.lowClipped; }
ISAAC_HD u32 rfRightClipped(const RealignFragment &f) { return (f.flags & 8) ? f.lowClipped : f.highClipped; }
// FragmentCollector::storeBclAndCigar (reverse alignments are stored reverse-complemented), then oligo::getUppercaseBaseFromBcl
ISAAC_HD char rfBase(const RealignFragment &f, u32 i)
{
    const bool reverse = f.flags & 8;
    const u8 b = reverse ? f.bcl[f.readLength - 1 - i] : f.bcl[i];
    if (!(b & 0xfc)) return 'N';
    const u32 code = reverse ? 3 - (b & 3) : (b & 3);
    return char(0x54474341u >> (8 * code));
}
struct RealignCigar { u32 words[RG_CIGAR_CAP]; u32 n; bool overflow; ISAAC_HD void push(u32 w) { if (n < RG_CIGAR_CAP) words[n++] = w; else overflow = true; } };
struct RealignIndex { u64 pos; const u32 *cigarBegin, *cigarEnd; };     // PackedFragmentBuffer::Index

ISAAC_HD bool rgIsMatch(char readBase, char referenceBase) { return readBase == 'n' || (readBase == referenceBase && referenceBase != 'N'); }

// RealignerGaps::findGaps (GapRealigner.cpp:99-145) into found[RG_FOUND_CAP]; returns the number of gaps, ~0u when there are more than the
// caller can use (more than `most` -- ten, or thirty with --realign-vigorously -- after the duplicates are gone)
ISAAC_HD u32 rgLowerBoundStart(const RealignGap *g, u32 lo, u32 hi, u64 pos, i32 length)
{ RealignGap key; key.pos = pos; key.length = length; while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (rgLess(g[mid], key)) lo = mid + 1; else hi = mid; } return lo; }
ISAAC_HD u32 rgLowerBoundEnd(const RealignGap *g, u32 lo, u32 hi, u64 endPos)
{ while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (rpPlus(g[mid].pos, rgLength(g[mid])) < endPos) lo = mid + 1; else hi = mid; } return lo; }
ISAAC_HD u32 rgFindGaps(const RealignerGapsView &v, u64 rangeBegin, u64 rangeEnd, RealignGap *found, u32 most = RG_MAX_GAPS_AT_A_TIME)
{
    const u32 s0 = rgLowerBoundStart(v.gaps, 0, v.nGaps, rangeBegin, -1000000), s1 = rgLowerBoundStart(v.gaps, s0, v.nGaps, rangeEnd, 0);
    // Gap(rangeBegin, 1).getDeletionEndPos() = rangeBegin + 1
    const u32 e0 = rgLowerBoundEnd(v.deletionEnds, 0, v.nDeletionEnds, rpPlus(rangeBegin, 1)), e1 = rgLowerBoundEnd(v.deletionEnds, e0, v.nDeletionEnds, rpPlus(rangeEnd, 1));
    const u32 nStarts = s1 - s0, nEnds = e1 - e0;
    if (nStarts > most || nEnds > most) return ~0u;          // already more distinct gaps than can be used
    u32 n = 0;
    for (u32 i = s0; i < s1; ++i) found[n++] = v.gaps[i];
    for (u32 i = e0; i < e1; ++i) found[n++] = v.deletionEnds[i];
    if (nStarts && nEnds)
    {   // consolidate: sort by (begin, length), drop duplicates
        for (u32 i = 1; i < n; ++i) { const RealignGap g = found[i]; u32 j = i; while (j && rgLess(g, found[j - 1])) { found[j] = found[j - 1]; --j; } found[j] = g; }
        u32 m = 0;
        for (u32 i = 0; i < n; ++i) if (!i || found[i].pos != found[m - 1].pos || found[i].length != found[m - 1].length) found[m++] = found[i];
        n = m;
    }
    return n;
}

// whether rgFindGaps would find anything at all in the range (the same four bounds, nothing copied): what decides, for nearly every fragment of a bin, that
// the realigner has nothing to do
ISAAC_HD bool rgAnyGap(const RealignerGapsView &v, u64 rangeBegin, u64 rangeEnd)
{
    const u32 s0 = rgLowerBoundStart(v.gaps, 0, v.nGaps, rangeBegin, -1000000), s1 = rgLowerBoundStart(v.gaps, s0, v.nGaps, rangeEnd, 0);
    if (s1 > s0) return true;
    const u32 e0 = rgLowerBoundEnd(v.deletionEnds, 0, v.nDeletionEnds, rpPlus(rangeBegin, 1)), e1 = rgLowerBoundEnd(v.deletionEnds, e0, v.nDeletionEnds, rpPlus(rangeEnd, 1));
    return e1 > e0;
}

// gapRealigner::OverlappingGapsFilter (OverlappingGapsFilter.hh:32-92, OverlappingGapsFilter.cpp:30-160) for at most thirty gaps (more: no choice at all, :42)
struct OverlapsFilter { u32 maxChoice, nOverlaps, overlaps[2 * RG_TRACKED_GAPS_MAX + 2]; };
ISAAC_HD void overlapsFilterInit(OverlapsFilter &f, const RealignGap *gaps, u32 nGaps)
{
    f.maxChoice = nGaps > RG_TRACKED_GAPS_MAX ? 0u : (1u << nGaps) - 1; f.nOverlaps = 0;
    if (!f.maxChoice) return;
    const u32 DELETION_END_INDEX_OFFSET = 0, DELETION_START_INDEX_OFFSET = 1024, INSERTION_INDEX_OFFSET = 2048;
    u32 endIndex[2 * RG_TRACKED_GAPS_MAX]; u64 endPos[2 * RG_TRACKED_GAPS_MAX]; u32 nEnds = 0;
    for (u32 i = 0; i < nGaps; ++i)
    {
        if (rgIsDeletion(gaps[i]))
        {
            endIndex[nEnds] = i + DELETION_START_INDEX_OFFSET; endPos[nEnds++] = gaps[i].pos;
            endIndex[nEnds] = i + DELETION_END_INDEX_OFFSET; endPos[nEnds++] = rgEndPos(gaps[i], false);
        }
        else { endIndex[nEnds] = i + INSERTION_INDEX_OFFSET; endPos[nEnds++] = rgEndPos(gaps[i], false); }
    }
    for (u32 i = 1; i < nEnds; ++i)      // orderByEndPosAndIndex: the keys are distinct, any sort gives the reference's order
    {
        const u32 ix = endIndex[i]; const u64 p = endPos[i]; u32 j = i;
        while (j && (p < endPos[j - 1] || (p == endPos[j - 1] && ix < endIndex[j - 1]))) { endIndex[j] = endIndex[j - 1]; endPos[j] = endPos[j - 1]; --j; }
        endIndex[j] = ix; endPos[j] = p;
    }
    u32 *ret = f.overlaps; u32 n = 0;
    u32 lastInsertionMask = 0; u64 lastInsertionPos = 0; u32 openDeletions = 0, openInsertions = 0;
    ret[n++] = 0;
    bool lastWasDeletionClose = true;
    for (u32 k = 0; k < nEnds; ++k)
    {
        if (DELETION_START_INDEX_OFFSET > endIndex[k])
        {
            const u32 gapMask = 1u << endIndex[k];
            if (lastWasDeletionClose) ret[n - 1] &= ~gapMask;
            else if (openDeletions + openInsertions > 1) { ret[n] = ret[n - 1] & ~lastInsertionMask & ~gapMask; ++n; lastInsertionMask = 0; openInsertions = 0; }
            else ret[n - 1] = 0;
            lastWasDeletionClose = true;
            --openDeletions;
        }
        else if (INSERTION_INDEX_OFFSET > endIndex[k])
        {
            const u32 gapMask = 1u << (endIndex[k] - DELETION_START_INDEX_OFFSET);
            if (lastInsertionMask && lastInsertionPos != endPos[k])
            {
                if (openDeletions + openInsertions > 1) { ret[n] = (ret[n - 1] & ~lastInsertionMask) | gapMask; ++n; }
                else ret[n - 1] = gapMask;
                lastInsertionMask = 0; openInsertions = 0;
            }
            else ret[n - 1] |= gapMask;
            ++openDeletions;
            lastWasDeletionClose = false;
        }
        else
        {
            const u32 gapMask = 1u << (endIndex[k] - INSERTION_INDEX_OFFSET);
            if (lastInsertionMask && lastInsertionPos != endPos[k])
            {
                if (openDeletions + openInsertions > 1) { ret[n] = (ret[n - 1] & ~lastInsertionMask) | gapMask; ++n; }
                else ret[n - 1] = gapMask;
                lastInsertionMask = gapMask; openInsertions = 1;
            }
            else { ret[n - 1] |= gapMask; lastInsertionMask |= gapMask; ++openInsertions; }
            lastInsertionPos = endPos[k];
            lastWasDeletionClose = false;
        }
    }
    if (openDeletions + openInsertions <= 1) --n;
    f.nOverlaps = n;
}
ISAAC_HD u32 rgPopcount(u32 v) { v = v - ((v >> 1) & 0x55555555u); v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u); return (((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24; }
ISAAC_HD u32 rgLsb(u32 v) { u32 n = 0; while (!((v >> n) & 1u)) ++n; return n; }
ISAAC_HD u32 overlapsFilterFind(const OverlapsFilter &f, u32 combination)
{
    for (u32 i = 0; i < f.nOverlaps; ++i) { const u32 both = combination & f.overlaps[i]; if (both && 1 < rgPopcount(both)) return both; }
    return 0;
}
ISAAC_HD u32 overlapsFilterNext(const OverlapsFilter &f, u32 combination)
{
    u32 increment = 1;
    while (combination < f.maxChoice)
    {
        combination += increment;
        const u32 overlapping = overlapsFilterFind(f, combination);
        if (!overlapping) return combination;
        increment = 1u << rgLsb(overlapping);
    }
    return 0;
}

struct RealignCtx { const DevReference *R; RealignParams P; };
ISAAC_HD const char *rgContig(const RealignCtx &x, u64 position) { return x.R->bases + x.R->contigOffset[refposContig(position)]; }
ISAAC_HD u64 rgContigLength(const RealignCtx &x, u64 position) { return contigLength(*x.R, refposContig(position)); }

// countMismatches (GapRealigner.cpp:235-259): read bases [firstBase, firstBase + length) against the reference from `pos` on
ISAAC_HD u32 rgCountMismatches(const RealignCtx &x, const RealignFragment &f, u32 firstBase, u64 pos, u32 length)
{
    const char *contig = rgContig(x, pos);
    const u64 at = refposPosition(pos), size = rgContigLength(x, pos);
    const u32 compareLength = u32(imin<u64>(length, at < size ? size - at : 0));
    u32 mismatches = 0;
    for (u32 i = 0; i < compareLength; ++i) mismatches += contig[at + i] != rfBase(f, firstBase + i);
    return mismatches;
}

ISAAC_HD u32 riBeginClippedLength(const RealignIndex &index) { return OP_SOFT_CLIP == cigarCode(*index.cigarBegin) ? cigarLen(*index.cigarBegin) : 0; }

struct RealignBounds { u64 beginPos, endPos; };
// GapRealigner::extractRealignmentBounds (:149-233): the first and one-past-the-last reference position the unclipped read touches
ISAAC_HD RealignBounds rgBounds(const RealignIndex &index)
{
    RealignBounds ret = { index.pos, index.pos };
    for (const u32 *it = index.cigarBegin; index.cigarEnd != it; ++it)
    {
        const u32 length = cigarLen(*it), code = cigarCode(*it);
        if (OP_ALIGN == code || OP_DELETE == code) ret.endPos = rpPlus(ret.endPos, length);
        else if (OP_SOFT_CLIP == code) { if (index.cigarBegin == it) ret.beginPos = rpPlus(ret.beginPos, -i64(length)); else ret.endPos = rpPlus(ret.endPos, length); }
    }
    return ret;
}

struct GapChoice { u32 editDistance, mismatches, cost, mappedLength; };

// GapRealigner::verifyGapsChoice (:505-651)
ISAAC_HD GapChoice rgVerifyGapsChoice(const RealignCtx &x, u32 choice, const RealignGap *gaps, u32 nGaps, u64 newBeginPos, const RealignFragment &fragment)
{
    GapChoice ret = { 0, 0, 0, 0 };
    const i32 rightClipped = i32(rfRightClipped(fragment));
    i32 basesLeft = fragment.readLength, leftClippedLeft = i32(rfLeftClipped(fragment));
    u64 lastGapEndPos = newBeginPos, lastGapBeginPos = 0;
    for (u32 k = 0; k < nGaps; ++k)
    {
        if (!(choice & (1u << k))) continue;
        const RealignGap &gap = gaps[k];
        if (rgEndPos(gap, true) <= lastGapEndPos || gap.pos < lastGapEndPos || gap.pos == lastGapBeginPos) { ret.cost = ~0u; return ret; }
        const i32 mappedBases = imin<i32>(basesLeft - rightClipped, i32(rpMinus(gap.pos, lastGapEndPos)));
        const u32 length = u32(mappedBases - imin(mappedBases, leftClippedLeft));
        const u32 mm = rgCountMismatches(x, fragment, u32(i32(fragment.readLength) - basesLeft + leftClippedLeft), rpPlus(lastGapEndPos, leftClippedLeft), length);
        ret.mappedLength += length; ret.editDistance += mm; ret.mismatches += mm; ret.cost += mm * x.P.mismatchCost;
        basesLeft -= mappedBases;
        leftClippedLeft -= imin(leftClippedLeft, mappedBases);
        u32 clippedGapLength = 0;
        if (rgIsInsertion(gap))
        {
            clippedGapLength = u32(imin<i32>(basesLeft - rightClipped, i32(rgLength(gap))));
            basesLeft -= i32(clippedGapLength);
            leftClippedLeft -= imin<i32>(leftClippedLeft, i32(rgLength(gap)));
        }
        else clippedGapLength = leftClippedLeft ? 0 : rgLength(gap);
        ret.editDistance += clippedGapLength;
        ret.cost += clippedGapLength ? (x.P.gapOpenCost + (clippedGapLength - 1) * x.P.gapExtendCost) : 0;
        lastGapEndPos = rgEndPos(gap, false);
        lastGapBeginPos = gap.pos;
        if (basesLeft == leftClippedLeft + rightClipped) break;
    }
    if (basesLeft > leftClippedLeft + rightClipped)
    {
        const u32 length = u32(basesLeft) - imin<u32>(u32(basesLeft), u32(leftClippedLeft)) - u32(rightClipped);
        const u64 firstUnclippedPos = rpPlus(lastGapEndPos, leftClippedLeft);
        if (refposPosition(firstUnclippedPos) > rgContigLength(x, firstUnclippedPos)) { ret.cost = ~0u; return ret; }
        const u32 mm = rgCountMismatches(x, fragment, u32(i32(fragment.readLength) - basesLeft + leftClippedLeft), firstUnclippedPos, length);
        ret.mappedLength += length; ret.editDistance += mm; ret.mismatches += mm; ret.cost += mm * x.P.mismatchCost;
    }
    return ret;
}

// GapRealigner::findStartPos (:842-968)
ISAAC_HD bool rgFindStartPos(u32 choice, const RealignGap *gaps, u64 binStartPos, u64 binEndPos, const RealignIndex &index, u32 pivotGapIndex, u64 pivotPos, u64 &ret)
{
    u64 lastGapEndPos = rpPlus(index.pos, -i64(riBeginClippedLength(index)));
    i64 offset = rpMinus(pivotPos, index.pos);
    for (const u32 *it = index.cigarBegin; index.cigarEnd != it; ++it)
    {
        if (lastGapEndPos > pivotPos) break;
        const u32 length = cigarLen(*it), code = cigarCode(*it);
        if (OP_ALIGN == code) lastGapEndPos = rpPlus(lastGapEndPos, length);
        else if (OP_INSERT == code) offset += length;
        else if (OP_DELETE == code)
        {
            lastGapEndPos = rpPlus(lastGapEndPos, length);
            if (lastGapEndPos > pivotPos) return false;
            offset -= length;
        }
        else { if (index.cigarBegin == it) offset += length; lastGapEndPos = rpPlus(lastGapEndPos, length); }       // OP_SOFT_CLIP
    }
    if (0 > offset) return false;
    u64 overlapPos = pivotPos;
    u32 basesLeft = u32(offset);
    for (u32 k = pivotGapIndex; k-- > 0;)
    {
        if (!(choice & (1u << k))) continue;
        const RealignGap &gap = gaps[k];
        if (rgEndPos(gap, false) > overlapPos) return false;
        if (rgIsInsertion(gap))
        {
            const u32 insertionBases = imin(basesLeft, rgLength(gap));
            offset -= insertionBases; basesLeft -= insertionBases;
            if (!basesLeft) break;
        }
        else { offset += rgLength(gap); overlapPos = gap.pos; }
    }
    if (rpPlus(binStartPos, offset) > pivotPos) return false;
    if (rpPlus(pivotPos, -offset) >= binEndPos) return false;
    ret = rpPlus(pivotPos, -offset);
    return true;
}

// GapRealigner::applyChoice (:660-833): the new CIGAR into `out`
ISAAC_HD bool rgApplyChoice(u32 choice, const RealignGap *gaps, u32 nGaps, u64 binEndPos, u64 contigEndPos, RealignIndex &index, const RealignFragment &fragment, RealignCigar &out)
{
    u64 newBeginPos = index.pos;
    out.n = 0;
    const i32 leftClipped = i32(rfLeftClipped(fragment)), rightClipped = i32(rfRightClipped(fragment));
    i32 basesLeft = fragment.readLength, leftClippedLeft = leftClipped, leftClippedInsertionBases = 0;
    if (leftClipped) out.push(cigarOp(u32(leftClipped), OP_SOFT_CLIP));
    u64 lastGapEndPos = newBeginPos;
    u32 lastOperation = 9;
    for (u32 k = 0; k < nGaps; ++k)
    {
        if (!(choice & (1u << k))) continue;
        const RealignGap &gap = gaps[k];
        const u64 gapClippedBeginPos = imax(gap.pos, newBeginPos);
        if (gapClippedBeginPos < lastGapEndPos) return false;          // "Overlapping gaps are not allowed": verifyGapsChoice never elects such a choice
        const i32 mappedBases = imin<i32>(basesLeft - rightClipped, i32(rpMinus(gapClippedBeginPos, lastGapEndPos)));
        const u32 softClippedMappedLength = u32(mappedBases - imin(mappedBases, leftClippedLeft));
        if (softClippedMappedLength) out.push(cigarOp(softClippedMappedLength, OP_ALIGN));
        basesLeft -= mappedBases;
        leftClippedLeft -= imin(mappedBases, leftClippedLeft);
        if (rgIsInsertion(gap))
        {
            const i32 clippedGapLength = imin<i32>(basesLeft - rightClipped, i32(rpMinus(rgEndPos(gap, true), gapClippedBeginPos)));
            const i32 softClippedGapLength = clippedGapLength - imin(clippedGapLength, leftClippedLeft);
            if (softClippedGapLength)
            {
                if (OP_INSERT == lastOperation && !mappedBases) out.words[out.n - 1] = cigarOp(cigarLen(out.words[out.n - 1]) + u32(softClippedGapLength), OP_INSERT);
                else { out.push(cigarOp(u32(softClippedGapLength), OP_INSERT)); lastOperation = OP_INSERT; }
            }
            basesLeft -= clippedGapLength;
            lastGapEndPos = gapClippedBeginPos;
            leftClippedLeft -= imin(clippedGapLength, leftClippedLeft);
            leftClippedInsertionBases += clippedGapLength - softClippedGapLength;
        }
        else
        {
            const i32 clippedGapLength = i32(rpMinus(rgEndPos(gap, true), gapClippedBeginPos));
            if (!leftClippedLeft)
            {
                if (OP_DELETE == lastOperation && !mappedBases) out.words[out.n - 1] = cigarOp(cigarLen(out.words[out.n - 1]) + u32(clippedGapLength), OP_DELETE);
                else { out.push(cigarOp(u32(clippedGapLength), OP_DELETE)); lastOperation = OP_DELETE; }
            }
            else newBeginPos = rpPlus(newBeginPos, clippedGapLength);
            lastGapEndPos = rgEndPos(gap, false);
        }
        if (basesLeft == leftClippedLeft + rightClipped) break;
    }
    if (basesLeft > leftClippedLeft + rightClipped)
    {
        const i32 basesToTheEndOfContig = i32(rpMinus(contigEndPos, lastGapEndPos)) - leftClippedLeft;
        const i32 mappedBases = imin(basesToTheEndOfContig, basesLeft - leftClippedLeft - rightClipped);
        if (mappedBases) out.push(cigarOp(u32(mappedBases), OP_ALIGN));
        basesLeft -= leftClippedLeft + mappedBases;
        leftClippedLeft = 0;
    }
    if (basesLeft) out.push(cigarOp(u32(basesLeft), OP_SOFT_CLIP));
    newBeginPos = rpPlus(newBeginPos, leftClipped - leftClippedInsertionBases);
    if (newBeginPos >= binEndPos) return false;
    index.pos = newBeginPos; index.cigarBegin = out.words; index.cigarEnd = out.words + out.n;
    return true;
}

// GapRealigner::compactCigar (:330-498): index.cigar -> out when gaps at the ends have to become soft clips
ISAAC_HD bool rgCompactCigar(const RealignCtx &x, u64 binEndPos, RealignIndex &index, RealignFragment &fragment, RealignCigar &out)
{
    const u32 *cigarIterator = index.cigarBegin;
    u32 softClipStart = 0; bool needCompacting = false;
    u64 newPos = index.pos;
    for (; index.cigarEnd != cigarIterator; ++cigarIterator)
    {
        const u32 length = cigarLen(*cigarIterator), code = cigarCode(*cigarIterator);
        if (OP_ALIGN == code) break;
        else if (OP_SOFT_CLIP == code) softClipStart += length;
        else if (OP_INSERT == code) { needCompacting = true; softClipStart += length; }
        else { needCompacting = true; if (binEndPos <= rpPlus(newPos, length)) return false; newPos = rpPlus(newPos, length); }       // OP_DELETE
    }
    if (index.cigarEnd == cigarIterator) return false;
    const u32 *cigarBackwardsIterator = index.cigarEnd - 1;
    u32 softClipEnd = 0;
    for (; cigarIterator != cigarBackwardsIterator; --cigarBackwardsIterator)
    {
        const u32 length = cigarLen(*cigarBackwardsIterator), code = cigarCode(*cigarBackwardsIterator);
        if (OP_ALIGN == code) break;
        else if (OP_SOFT_CLIP == code) softClipEnd += length;
        else if (OP_INSERT == code) { needCompacting = true; softClipEnd += length; }
        else needCompacting = true;
    }
    const u32 *middleBegin = cigarIterator, *middleEnd = cigarBackwardsIterator + 1;
    // edit distance and observed length over the middle, which stays where it is until it has been read
    u32 newEditDistance = 0, firstBase = softClipStart;
    const u64 startPos = needCompacting ? newPos : index.pos;
    u64 newEndPos = startPos;
    for (const u32 *it = middleBegin; it != middleEnd; ++it)
    {
        const u32 length = cigarLen(*it), code = cigarCode(*it);
        if (OP_ALIGN == code) { newEditDistance += rgCountMismatches(x, fragment, firstBase, newEndPos, length); newEndPos = rpPlus(newEndPos, length); firstBase += length; }
        else if (OP_INSERT == code) { newEditDistance += length; firstBase += length; }
        else if (OP_DELETE == code) { newEditDistance += length; newEndPos = rpPlus(newEndPos, length); }
    }
    if (needCompacting)
    {
        out.n = 0;
        if (softClipStart) out.push(cigarOp(softClipStart, OP_SOFT_CLIP));
        for (const u32 *it = middleBegin; it != middleEnd; ++it) out.push(*it);
        if (softClipEnd) out.push(cigarOp(softClipEnd, OP_SOFT_CLIP));
        index.cigarBegin = out.words; index.cigarEnd = out.words + out.n; index.pos = newPos;
    }
    fragment.editDistance = u16(newEditDistance);
    fragment.fStrandPosition = index.pos;
    fragment.observedLength = u32(rpMinus(newEndPos, fragment.fStrandPosition));
    return true;
}

// alignment::clipMismatches<5> (Alignment.hh:55-88) walking the read from firstBase in direction step and the contig from refAt likewise
ISAAC_HD void rgClipMismatches(const RealignCtx &x, const RealignFragment &f, i32 firstBase, i32 step, u32 sequenceLength, const char *contig, i64 refAt, u32 referenceLength, u32 &moved, u32 &editDistanceAdjustment)
{
    const u32 CONSECUTIVE_MATCHES_MIN = 5;
    u32 matchesInARow = 0, mismatches = 0, mismatchesUnclipped = 0, ret = 0;
    while (ret != sequenceLength && ret != referenceLength && CONSECUTIVE_MATCHES_MIN > matchesInARow)
    {
        const char sequenceBase = rfBase(f, u32(firstBase + step * i32(ret))), referenceBase = contig[refAt + i64(step) * i64(ret)];
        if (rgIsMatch(sequenceBase, referenceBase)) { ++matchesInARow; mismatchesUnclipped += (sequenceBase != referenceBase); }
        else { matchesInARow = 0; mismatchesUnclipped = 0; }
        mismatches += (sequenceBase != referenceBase);
        ++ret;
    }
    (void)x;
    if (CONSECUTIVE_MATCHES_MIN == matchesInARow) { moved = ret - matchesInARow; editDistanceAdjustment = mismatches - mismatchesUnclipped; } else { moved = 0; editDistanceAdjustment = 0; }
}
// build::SemialignedEndsClipper::clipLeftSide / clipRightSide (lib/build/SemialignedEndsClipper.cpp:34-140)
ISAAC_HD bool rgClipLeftSide(const RealignCtx &x, u64 binEndPos, RealignIndex &index, RealignFragment &fragment, RealignCigar &out)
{
    const u32 *oldCigarBegin = index.cigarBegin;
    u32 length = cigarLen(*oldCigarBegin), code = cigarCode(*oldCigarBegin), softClippedBeginBases = 0, firstBase = 0;
    if (OP_SOFT_CLIP == code) { ++oldCigarBegin; softClippedBeginBases = length; firstBase = length; length = cigarLen(*oldCigarBegin); code = cigarCode(*oldCigarBegin); }
    if (OP_ALIGN != code) return false;
    u32 mappedBeginBases = length, moved, adjustment;
    const u64 at = refposPosition(index.pos);
    rgClipMismatches(x, fragment, i32(firstBase), 1, mappedBeginBases, rgContig(x, index.pos), i64(at), u32(rgContigLength(x, index.pos) - at), moved, adjustment);
    if (!moved || !(rpPlus(index.pos, moved) < binEndPos)) return false;
    softClippedBeginBases += moved; mappedBeginBases -= moved;
    index.pos = rpPlus(index.pos, moved);
    fragment.fStrandPosition = rpPlus(fragment.fStrandPosition, moved);
    fragment.observedLength -= moved; fragment.editDistance = u16(fragment.editDistance - adjustment);
    out.n = 0;
    out.push(cigarOp(softClippedBeginBases, OP_SOFT_CLIP)); out.push(cigarOp(mappedBeginBases, OP_ALIGN));
    for (const u32 *it = oldCigarBegin + 1; it != index.cigarEnd; ++it) out.push(*it);
    index.cigarBegin = out.words; index.cigarEnd = out.words + out.n;
    return true;
}
ISAAC_HD bool rgClipRightSide(const RealignCtx &x, RealignIndex &index, RealignFragment &fragment, RealignCigar &out)
{
    const u32 *oldCigarEnd = index.cigarEnd;
    u32 length = cigarLen(*(oldCigarEnd - 1)), code = cigarCode(*(oldCigarEnd - 1)), softClippedEndBases = 0, skipped = 0;
    if (OP_SOFT_CLIP == code) { --oldCigarEnd; softClippedEndBases = length; skipped = length; length = cigarLen(*(oldCigarEnd - 1)); code = cigarCode(*(oldCigarEnd - 1)); }
    if (OP_ALIGN != code) return false;
    u32 mappedEndBases = length, moved, adjustment;
    const u64 referenceEnd = refposPosition(index.pos) + fragment.observedLength;
    rgClipMismatches(x, fragment, i32(fragment.readLength) - 1 - i32(skipped), -1, mappedEndBases, rgContig(x, index.pos), i64(referenceEnd) - 1, u32(referenceEnd), moved, adjustment);
    if (!moved) return false;
    softClippedEndBases += moved; mappedEndBases -= moved;
    fragment.observedLength -= moved; fragment.editDistance = u16(fragment.editDistance - adjustment);
    out.n = 0;
    for (const u32 *it = index.cigarBegin; it != oldCigarEnd - 1; ++it) out.push(*it);
    out.push(cigarOp(mappedEndBases, OP_ALIGN)); out.push(cigarOp(softClippedEndBases, OP_SOFT_CLIP));
    index.cigarBegin = out.words; index.cigarEnd = out.words + out.n;
    return true;
}

// GapRealigner::getAlignmentCost (:1020-1040)
ISAAC_HD u32 rgAlignmentCost(const RealignCtx &x, const RealignFragment &fragment, const RealignIndex &index, u32 &editDistance, i32 &mismatchesPercent)
{
    u32 gapsCount = 0, mappedLength = 0; u16 totalGapsLength = 0;
    for (const u32 *it = index.cigarBegin; it != index.cigarEnd; ++it)
    {
        const u32 length = cigarLen(*it), code = cigarCode(*it);
        if (OP_ALIGN == code) mappedLength += length;
        else if (OP_INSERT == code || OP_DELETE == code) { totalGapsLength = u16(totalGapsLength + length); ++gapsCount; }
    }
    editDistance = fragment.editDistance;
    const u32 mismatches = u32(fragment.editDistance) - totalGapsLength;
    mismatchesPercent = i32(mismatches * 100 / mappedLength);
    return mismatches * x.P.mismatchCost + gapsCount * x.P.gapOpenCost + x.P.gapExtendCost * (totalGapsLength - gapsCount);
}

// One turn of the loop of GapRealigner::realign (:1073-1262) without updatePairDetails (the pair's fields are brought up to date by
// realignPairDetails once both ends are final: the turns of one fragment's loop do not look at them).  `index` comes in pointing at the fragment's CIGAR -- its own,
// or `result` after an earlier turn --; on true it points into `result` and fragment.fStrandPosition / observedLength / editDistance are the new ones.
ISAAC_HD bool realignOnce(const RealignCtx &x, const RealignerGapsView &gapsView, u64 binStartPos, u64 binEndPos, RealignIndex &index, RealignFragment &fragment, RealignCigar &result)
{
    if (fragment.flags & 2) return false;
    binEndPos = refpos(refposContig(binEndPos), imin<u64>(refposPosition(binEndPos), rgContigLength(x, binEndPos)));
    const u16 DODGY = 0xffff;
    if (!(fragment.editDistance &&
          (!(fragment.flags & 1) || (!(fragment.flags & 4) && binStartPos <= fragment.mateFStrandPosition && binEndPos > fragment.mateFStrandPosition)) &&
          (x.P.realignDodgyFragments || DODGY != fragment.alignmentScore || DODGY != fragment.templateAlignmentScore) &&
          (refposPosition(index.pos) >= riBeginClippedLength(index)))) return false;
    index.pos = fragment.fStrandPosition;
    const RealignBounds bounds = rgBounds(index);
    RealignGap gaps[RG_FOUND_CAP];
    const bool vigorous = 0 != x.P.realignGapsVigorously;
    const u32 nGaps = rgFindGaps(gapsView, bounds.beginPos, bounds.endPos, gaps, vigorous ? RG_TRACKED_GAPS_MAX : RG_MAX_GAPS_AT_A_TIME);
    if (~0u == nGaps || !nGaps) return false;
    if (!vigorous && RG_MAX_GAPS_AT_A_TIME < nGaps) return false;                  // :1117-1121
    if (RG_TRACKED_GAPS_MAX < nGaps) return false;                                  // OverlappingGapsFilter.hh:42: no choice to look at
    OverlapsFilter filter;
    overlapsFilterInit(filter, gaps, nGaps);
    u32 bestEditDistance = 0; i32 originalMismatchesPercent = 0;
    u32 bestCost = rgAlignmentCost(x, fragment, index, bestEditDistance, originalMismatchesPercent);
    u64 bestStartPos = index.pos;
    u32 bestChoice = 0, evaluatedSoFar = 0;
    for (u32 choice = 0; (choice = overlapsFilterNext(filter, choice));)
    {
        if (((1u << RG_MAX_GAPS_AT_A_TIME) - 1) < evaluatedSoFar++) break;
        for (u32 pivotGapIndex = 0; pivotGapIndex < nGaps; ++pivotGapIndex)
        {
            if (!(choice & (1u << pivotGapIndex))) continue;
            const RealignGap &pivotGap = gaps[pivotGapIndex];
            for (u32 after = 0; after < 2; ++after)
            {
                u64 newStartPos;
                if (!after && !(pivotGap.pos >= binStartPos)) continue;
                if (!rgFindStartPos(choice, gaps, binStartPos, binEndPos, index, pivotGapIndex + after, after ? rgEndPos(pivotGap, false) : pivotGap.pos, newStartPos)) continue;
                const GapChoice c = rgVerifyGapsChoice(x, choice, gaps, nGaps, newStartPos, fragment);
                if (c.mappedLength && (c.cost < bestCost || (c.cost == bestCost && c.editDistance < bestEditDistance)) && i32(c.mismatches * 100 / c.mappedLength) <= originalMismatchesPercent)
                { bestEditDistance = c.editDistance; bestChoice = choice; bestStartPos = newStartPos; bestCost = c.cost; }
            }
        }
    }
    if (!bestChoice || !(binEndPos > bestStartPos)) return false;
    RealignIndex tmp = index; tmp.pos = bestStartPos;
    RealignFragment changed = fragment;
    RealignCigar a, b; a.n = b.n = 0; a.overflow = b.overflow = false;
    const u64 contigEndPos = refpos(refposContig(binEndPos), rgContigLength(x, binEndPos));
    if (!rgApplyChoice(bestChoice, gaps, nGaps, binEndPos, contigEndPos, tmp, changed, a)) return false;
    if (!rgCompactCigar(x, binEndPos, tmp, changed, b)) return false;
    if (x.P.clipSemialigned)
    {   // SemialignedEndsClipper::clip: the left side into whichever buffer the CIGAR is not in, then the right side likewise
        RealignCigar &free1 = (tmp.cigarBegin == a.words) ? b : a;
        const bool left = rgClipLeftSide(x, binEndPos, tmp, changed, free1);
        RealignCigar &free2 = (tmp.cigarBegin == a.words) ? b : a;
        (void)left;
        rgClipRightSide(x, tmp, changed, free2);
    }
    if (a.overflow || b.overflow) return false;        // a CIGAR beyond RG_CIGAR_CAP operations: cannot happen with ten gaps, checked all the same
    result.n = 0; result.overflow = false;
    for (const u32 *it = tmp.cigarBegin; it != tmp.cigarEnd; ++it) result.push(*it);
    index.pos = tmp.pos; index.cigarBegin = result.words; index.cigarEnd = result.words + result.n;
    fragment = changed;
    return true;
}
// GapRealigner::realign (:1053-1268): once, or with --realign-vigorously again and again while a turn finds a better alignment (:1241).  True: the fragment was realigned.
ISAAC_HD bool realignFragment(const RealignCtx &x, const RealignerGapsView &gapsView, u64 binStartPos, u64 binEndPos, RealignIndex &index, RealignFragment &fragment, RealignCigar &result)
{
    bool any = false;
    while (realignOnce(x, gapsView, binStartPos, binEndPos, index, fragment, result))
    {
        any = true;
        if (!x.P.realignGapsVigorously) break;
    }
    return any;
}

// GapRealigner::updatePairDetails (:267-318) for a pair whose ends are both final; `f` is the end realigned last (the reference runs the
// reverse-strand ends and shadows of a bin before its forward-strand ends: BinSorter::resolveDuplicates fills the index in that order)
struct PairEnd { u64 fStrandPosition, mateFStrandPosition; i32 bamTlen; u32 observedLength, flags; };
ISAAC_HD i32 rgTlen(u64 fragmentBeginPos, u64 fragmentEndPos, u64 mateBeginPos, u64 mateEndPos, bool firstRead)      // io::FragmentHeader::getTlen (Fragment.hh:199-212)
{
    const u64 lo = imin(fragmentBeginPos, mateBeginPos), hi = imax(fragmentEndPos, mateEndPos);
    const u64 distance = ((hi >> 1) - (u64(1) << 40)) - ((lo >> 1) - (u64(1) << 40));                                   // getLocation differences
    const i64 ret = fragmentBeginPos < mateBeginPos ? i64(distance) : (fragmentBeginPos > mateBeginPos || !firstRead) ? -i64(distance) : i64(distance);
    return i32(ret);
}

} // namespace isaac
