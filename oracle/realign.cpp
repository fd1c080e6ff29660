// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  CPU restatement of the gap realigner of the reference's BAM stage
// (--realign-gaps sample|project|all):
//   gapRealigner::Gap, RealignerGaps::addGaps / finalizeGaps / findGaps   include/build/gapRealigner/Gap.hh:31-78, include/build/GapRealigner.hh:37-128,
//                                                                          lib/build/GapRealigner.cpp:46-145
//   gapRealigner::OverlappingGapsFilter                                    include/build/gapRealigner/OverlappingGapsFilter.hh:32-92,
//                                                                          lib/build/gapRealigner/OverlappingGapsFilter.cpp:30-160
//   GapRealigner::realign and everything it calls                          lib/build/GapRealigner.cpp:149-1292
//   build::SemialignedEndsClipper                                          lib/build/SemialignedEndsClipper.cpp:34-156 (+ alignment::clipMismatches, Alignment.hh:55-88)
// Pinned by the literal cases of lib/build/cppunit/testGapRealigner.cpp (tests/golden/gap_realigner.json).
#include "oracle.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace oracle
{

namespace
{
// ReferencePosition arithmetic (ReferencePosition.hh:139-172)
inline ReferencePosition plus(ReferencePosition p, long offset) { return ReferencePosition::fromValue(p.value + (uint64_t(offset) << 1)); }
inline long minus(ReferencePosition l, ReferencePosition r) { return long(l.getPosition()) - long(r.getPosition()); }
inline bool lessEq(ReferencePosition l, ReferencePosition r) { return !(r < l); }

enum { OP_ALIGN = 0, OP_INSERT = 1, OP_DELETE = 2, OP_SOFT_CLIP = 4 };
inline unsigned opLength(uint32_t w) { return w >> 4; }
inline unsigned opCode(uint32_t w) { return w & 0xf; }
inline uint32_t encode(unsigned length, unsigned code) { return (length << 4) | code; }

inline char uppercaseBaseFromBcl(unsigned char b) { return !(b & 0xfc) ? 'N' : "ACGT"[b & 3]; }     // oligo::getUppercaseBaseFromBcl
inline bool isMatch(char readBase, char referenceBase) { return readBase == 'n' || (readBase == referenceBase && referenceBase != 'N'); }

bool orderByGapStartAndTypeLength(const RealignGap &l, const RealignGap &r) { return l.pos < r.pos || (l.pos == r.pos && l.length < r.length); }
bool orderByDeletionGapEnd(const RealignGap &l, const RealignGap &r) { return l.deletionEndPos() < r.deletionEndPos(); }
bool gapEqual(const RealignGap &l, const RealignGap &r) { return l.pos == r.pos && l.length == r.length; }

unsigned countBitsSet(unsigned v) { return unsigned(__builtin_popcount(v)); }
int lsbSet(unsigned v) { return __builtin_ctz(v); }
} // namespace

ReferencePosition RealignGap::endPos(bool fatInsertions) const { return (isDeletion() || fatInsertions) ? plus(pos, std::abs(length)) : pos; }
ReferencePosition RealignGap::deletionEndPos() const { return plus(pos, std::abs(length)); }

// RealignerGaps::addGaps (GapRealigner.hh:54-104)
void RealignerGaps::addGaps(ReferencePosition fStrandPosition, const uint32_t *cigarBegin, const uint32_t *cigarEnd)
{
    ReferencePosition pos = fStrandPosition;
    for (const uint32_t *it = cigarBegin; cigarEnd != it; ++it)
    {
        const unsigned length = opLength(*it), code = opCode(*it);
        if (OP_ALIGN == code) pos = plus(pos, length);
        else if (OP_INSERT == code) gapGroups.push_back(RealignGap(pos, -int(length)));
        else if (OP_DELETE == code) { gapGroups.push_back(RealignGap(pos, int(length))); pos = plus(pos, length); }
        else if (OP_SOFT_CLIP != code) throw std::logic_error("Unexpected Cigar OpCode");
    }
}
// RealignerGaps::finalizeGaps (GapRealigner.cpp:86-94)
void RealignerGaps::finalizeGaps()
{
    std::sort(gapGroups.begin(), gapGroups.end(), orderByGapStartAndTypeLength);
    gapGroups.erase(std::unique(gapGroups.begin(), gapGroups.end(), gapEqual), gapGroups.end());
    deletionEndGroups.clear();
    for (const RealignGap &g : gapGroups) if (g.isDeletion()) deletionEndGroups.push_back(g);
    std::sort(deletionEndGroups.begin(), deletionEndGroups.end(), orderByDeletionGapEnd);
}
// RealignerGaps::findGaps (GapRealigner.cpp:99-145): foundGaps.capacity() of the reference's buffer is MAX_GAPS_AT_A_TIME * 10 in GapRealigner
// (the test hands in 100000)
void RealignerGaps::findGaps(ReferencePosition rangeBegin, ReferencePosition rangeEnd, std::vector<RealignGap> &foundGaps, size_t capacity) const
{
    foundGaps.clear();
    const std::vector<RealignGap>::const_iterator startsFirst = std::lower_bound(gapGroups.begin(), gapGroups.end(), RealignGap(rangeBegin, -1000000), orderByGapStartAndTypeLength);
    const std::vector<RealignGap>::const_iterator startsSecond = std::lower_bound(startsFirst, gapGroups.end(), RealignGap(rangeEnd, 0), orderByGapStartAndTypeLength);
    const std::vector<RealignGap>::const_iterator endsFirst = std::lower_bound(deletionEndGroups.begin(), deletionEndGroups.end(), RealignGap(rangeBegin, 1), orderByDeletionGapEnd);
    const std::vector<RealignGap>::const_iterator endsSecond = std::lower_bound(endsFirst, deletionEndGroups.end(), RealignGap(rangeEnd, 1), orderByDeletionGapEnd);
    if (capacity < size_t(startsSecond - startsFirst) + size_t(endsSecond - endsFirst)) return;      // "Too many gaps": none
    foundGaps.insert(foundGaps.end(), startsFirst, startsSecond);
    foundGaps.insert(foundGaps.end(), endsFirst, endsSecond);
    if (endsFirst != endsSecond && startsFirst != startsSecond)
    {
        std::sort(foundGaps.begin(), foundGaps.end(), orderByGapStartAndTypeLength);
        foundGaps.erase(std::unique(foundGaps.begin(), foundGaps.end(), gapEqual), foundGaps.end());
    }
}

// ---- OverlappingGapsFilter
OverlappingGapsFilter::OverlappingGapsFilter(const std::vector<RealignGap> &gaps)
    : maxChoice(gaps.size() > MAX_TRACKED_DELETIONS ? 0 : (1u << gaps.size()) - 1)
{
    if (maxChoice) findOverlaps(gaps);
}
void OverlappingGapsFilter::findOverlaps(const std::vector<RealignGap> &gaps)
{
    const unsigned DELETION_END_INDEX_OFFSET = 0, DELETION_START_INDEX_OFFSET = 1024, INSERTION_INDEX_OFFSET = 2048;
    typedef std::pair<unsigned, ReferencePosition> GapEnd;
    std::vector<GapEnd> gapEnds;
    int gapIndex = 0;
    for (std::vector<RealignGap>::const_iterator it = gaps.begin(); gaps.end() != it; ++it, ++gapIndex)
    {
        if (it->isDeletion())
        {
            gapEnds.push_back(GapEnd(gapIndex + DELETION_START_INDEX_OFFSET, it->pos));
            gapEnds.push_back(GapEnd(gapIndex + DELETION_END_INDEX_OFFSET, it->endPos(false)));
        }
        else gapEnds.push_back(GapEnd(gapIndex + INSERTION_INDEX_OFFSET, it->endPos(false)));
    }
    std::sort(gapEnds.begin(), gapEnds.end(), [](const GapEnd &l, const GapEnd &r) { return l.second < r.second || (l.second == r.second && l.first < r.first); });
    std::vector<unsigned> &ret = overlappingGaps;
    unsigned lastInsertionMask = 0;
    ReferencePosition lastInsertionPos;
    unsigned openDeletions = 0, openInsertions = 0;
    ret.push_back(0);
    bool lastWasDeletionClose = true;
    for (const GapEnd &gapEnd : gapEnds)
    {
        if (DELETION_START_INDEX_OFFSET > gapEnd.first)
        {   // close deletion
            const unsigned gapMask = 1u << gapEnd.first;
            if (lastWasDeletionClose) ret.back() &= ~gapMask;
            else if (openDeletions + openInsertions > 1)
            {
                ret.push_back(ret.back() & ~lastInsertionMask & ~gapMask);
                lastInsertionMask = 0; openInsertions = 0;
            }
            else ret.back() = 0;
            lastWasDeletionClose = true;
            --openDeletions;
        }
        else if (INSERTION_INDEX_OFFSET > gapEnd.first)
        {   // open deletion
            const unsigned gapMask = 1u << (gapEnd.first - DELETION_START_INDEX_OFFSET);
            if (lastInsertionMask && lastInsertionPos != gapEnd.second)
            {
                if (openDeletions + openInsertions > 1) ret.push_back((ret.back() & ~lastInsertionMask) | gapMask);
                else ret.back() = gapMask;
                lastInsertionMask = 0; openInsertions = 0;
            }
            else ret.back() |= gapMask;
            ++openDeletions;
            lastWasDeletionClose = false;
        }
        else
        {   // insertion
            const unsigned gapMask = 1u << (gapEnd.first - INSERTION_INDEX_OFFSET);
            if (lastInsertionMask && lastInsertionPos != gapEnd.second)
            {
                if (openDeletions + openInsertions > 1) ret.push_back((ret.back() & ~lastInsertionMask) | gapMask);
                else ret.back() = gapMask;
                lastInsertionMask = gapMask; openInsertions = 1;
            }
            else { ret.back() |= gapMask; lastInsertionMask |= gapMask; ++openInsertions; }
            lastInsertionPos = gapEnd.second;
            lastWasDeletionClose = false;
        }
        if (ret.size() > MAX_TRACKED_OVERLAPS) throw std::logic_error("OverlappingGapsFilter: more overlaps than the reference's FiniteCapacityVector holds");
    }
    if (openDeletions + openInsertions <= 1) ret.pop_back();
}
unsigned OverlappingGapsFilter::findOverlaps(unsigned combination) const
{
    for (const unsigned overlap : overlappingGaps)
    {
        const unsigned both = combination & overlap;
        if (both && 1 < countBitsSet(both)) return both;
    }
    return 0;
}
unsigned OverlappingGapsFilter::next(unsigned combination) const
{
    unsigned increment = 1;
    while (combination < maxChoice)
    {
        combination += increment;
        const unsigned overlapping = findOverlaps(combination);
        if (!overlapping) return combination;
        increment = 1u << lsbSet(overlapping);
    }
    return 0;
}

// ---- GapRealigner
namespace
{
struct RealignmentBounds { ReferencePosition beginPos, firstGapStartPos, lastGapEndPos, endPos; };
struct GapChoice { unsigned editDistance = 0, mismatches = 0, cost = 0, mappedLength = 0; };

unsigned beginClippedLength(const RealignIndex &index) { return OP_SOFT_CLIP == opCode(*index.cigarBegin) ? opLength(*index.cigarBegin) : 0; }
ReferencePosition unclippedPosition(const RealignIndex &index) { return plus(index.pos, -long(beginClippedLength(index))); }

// GapRealigner::extractRealignmentBounds (:149-233)
RealignmentBounds extractRealignmentBounds(const RealignIndex &index)
{
    RealignmentBounds ret = { index.pos, index.pos, index.pos, index.pos };
    const uint32_t *it = index.cigarBegin;
    for (; index.cigarEnd != it; ++it)
    {
        const unsigned length = opLength(*it), code = opCode(*it);
        if (OP_ALIGN == code) { ret.firstGapStartPos = plus(ret.firstGapStartPos, length); ret.endPos = plus(ret.endPos, length); }
        else if (OP_INSERT == code) { ret.lastGapEndPos = ret.endPos; ++it; break; }
        else if (OP_DELETE == code) { ret.endPos = plus(ret.endPos, length); ret.lastGapEndPos = ret.endPos; ++it; break; }
        else if (OP_SOFT_CLIP == code)
        {
            if (index.cigarBegin == it) { ret.beginPos = plus(ret.beginPos, -long(length)); ret.lastGapEndPos = ret.firstGapStartPos = ret.beginPos; }
            else ret.endPos = plus(ret.endPos, length);
        }
        else throw std::logic_error("Unexpected Cigar OpCode");
    }
    for (; index.cigarEnd != it; ++it)
    {
        const unsigned length = opLength(*it), code = opCode(*it);
        if (OP_ALIGN == code) ret.endPos = plus(ret.endPos, length);
        else if (OP_INSERT == code) ret.lastGapEndPos = ret.endPos;
        else if (OP_DELETE == code) { ret.endPos = plus(ret.endPos, length); ret.lastGapEndPos = ret.endPos; }
        else if (OP_SOFT_CLIP == code) ret.endPos = plus(ret.endPos, length);
        else throw std::logic_error("Unexpected Cigar OpCode");
    }
    return ret;
}

// countMismatches (:235-259)
unsigned countMismatches(const ContigList &reference, const unsigned char *bases, ReferencePosition pos, unsigned length)
{
    const std::vector<char> &forward = reference.at(pos.getContigId()).forward;
    const size_t at = pos.getPosition();
    const unsigned compareLength = unsigned(std::min<size_t>(length, at < forward.size() ? forward.size() - at : 0));
    unsigned mismatches = 0;
    for (unsigned i = 0; i < compareLength; ++i) mismatches += forward[at + i] != uppercaseBaseFromBcl(bases[i]);
    return mismatches;
}

// alignment::clipMismatches<5> (Alignment.hh:55-88) over forward or reverse iteration
template <typename SeqAt, typename RefAt>
std::pair<unsigned, unsigned> clipMismatches(unsigned sequenceLength, SeqAt sequenceAt, unsigned referenceLength, RefAt referenceAt)
{
    const unsigned CONSECUTIVE_MATCHES_MIN = 5;
    unsigned matchesInARow = 0, editDistanceMismatches = 0, editDistanceMismatchesUnclipped = 0, ret = 0;
    while (ret != sequenceLength && ret != referenceLength && CONSECUTIVE_MATCHES_MIN > matchesInARow)
    {
        const char sequenceBase = uppercaseBaseFromBcl(sequenceAt(ret)), referenceBase = referenceAt(ret);
        if (isMatch(sequenceBase, referenceBase)) { ++matchesInARow; editDistanceMismatchesUnclipped += (sequenceBase != referenceBase); }
        else { matchesInARow = 0; editDistanceMismatchesUnclipped = 0; }
        editDistanceMismatches += (sequenceBase != referenceBase);
        ++ret;
    }
    return CONSECUTIVE_MATCHES_MIN == matchesInARow ? std::make_pair(ret - matchesInARow, editDistanceMismatches - editDistanceMismatchesUnclipped) : std::make_pair(0u, 0u);
}
} // namespace

struct GapRealigner::Impl
{
    const GapRealigner &self;
    const ContigList &reference;
    std::vector<uint32_t> &realignedCigars;

    // build::SemialignedEndsClipper::clipLeftSide / clipRightSide / clip (SemialignedEndsClipper.cpp:34-156)
    bool clipLeftSide(ReferencePosition binEndPos, RealignIndex &index, RealignFragment &fragment)
    {
        const unsigned char *sequenceBegin = fragment.bases;
        const uint32_t *oldCigarBegin = index.cigarBegin;
        unsigned length = opLength(*oldCigarBegin), code = opCode(*oldCigarBegin), softClippedBeginBases = 0;
        if (OP_SOFT_CLIP == code) { ++oldCigarBegin; softClippedBeginBases = length; sequenceBegin += length; length = opLength(*oldCigarBegin); code = opCode(*oldCigarBegin); }
        if (OP_ALIGN == code)
        {
            unsigned mappedBeginBases = length;
            const std::vector<char> &forward = reference.at(index.pos.getContigId()).forward;
            const size_t at = index.pos.getPosition();
            const std::pair<unsigned, unsigned> clipped = clipMismatches(mappedBeginBases, [&](unsigned i) { return sequenceBegin[i]; },
                                                                         unsigned(forward.size() - at), [&](unsigned i) { return forward[at + i]; });
            if (clipped.first && plus(index.pos, clipped.first) < binEndPos)
            {
                softClippedBeginBases += clipped.first; mappedBeginBases -= clipped.first;
                index.pos = plus(index.pos, clipped.first);
                fragment.fStrandPosition = plus(fragment.fStrandPosition, clipped.first);
                fragment.observedLength -= clipped.first; fragment.editDistance -= clipped.second;
                const std::vector<uint32_t> tail(oldCigarBegin + 1, index.cigarEnd);
                const size_t before = realignedCigars.size();
                realignedCigars.push_back(encode(softClippedBeginBases, OP_SOFT_CLIP));
                realignedCigars.push_back(encode(mappedBeginBases, OP_ALIGN));
                realignedCigars.insert(realignedCigars.end(), tail.begin(), tail.end());
                index.cigarBegin = &realignedCigars.at(before); index.cigarEnd = &realignedCigars.back() + 1;
                return true;
            }
        }
        return false;
    }
    bool clipRightSide(RealignIndex &index, RealignFragment &fragment)
    {
        const unsigned char *sequenceEnd = fragment.bases + fragment.readLength;
        const uint32_t *oldCigarEnd = index.cigarEnd;
        unsigned length = opLength(*(oldCigarEnd - 1)), code = opCode(*(oldCigarEnd - 1)), softClippedEndBases = 0, skipped = 0;
        if (OP_SOFT_CLIP == code) { --oldCigarEnd; softClippedEndBases = length; skipped = length; length = opLength(*(oldCigarEnd - 1)); code = opCode(*(oldCigarEnd - 1)); }
        if (OP_ALIGN == code)
        {
            unsigned mappedEndBases = length;
            const std::vector<char> &forward = reference.at(index.pos.getContigId()).forward;
            const size_t referenceEnd = index.pos.getPosition() + fragment.observedLength;      // one past the last reference base of the alignment
            const std::pair<unsigned, unsigned> clipped = clipMismatches(mappedEndBases, [&](unsigned i) { return *(sequenceEnd - 1 - skipped - i); },
                                                                         unsigned(referenceEnd), [&](unsigned i) { return forward[referenceEnd - 1 - i]; });
            if (clipped.first)
            {
                softClippedEndBases += clipped.first; mappedEndBases -= clipped.first;
                fragment.observedLength -= clipped.first; fragment.editDistance -= clipped.second;
                const std::vector<uint32_t> head(index.cigarBegin, oldCigarEnd - 1);
                const size_t before = realignedCigars.size();
                realignedCigars.insert(realignedCigars.end(), head.begin(), head.end());
                realignedCigars.push_back(encode(mappedEndBases, OP_ALIGN));
                realignedCigars.push_back(encode(softClippedEndBases, OP_SOFT_CLIP));
                index.cigarBegin = &realignedCigars.at(before); index.cigarEnd = &realignedCigars.back() + 1;
                return true;
            }
        }
        return false;
    }

    // GapRealigner::compactCigar (:330-498)
    bool compactCigar(ReferencePosition binEndPos, RealignIndex &index, RealignFragment &fragment)
    {
        const uint32_t *cigarIterator = index.cigarBegin;
        unsigned softClipStart = 0;
        bool needCompacting = false;
        ReferencePosition newPos = index.pos;
        for (; index.cigarEnd != cigarIterator; ++cigarIterator)
        {
            const unsigned length = opLength(*cigarIterator), code = opCode(*cigarIterator);
            if (OP_ALIGN == code) break;
            else if (OP_SOFT_CLIP == code) softClipStart += length;
            else if (OP_INSERT == code) { needCompacting = true; softClipStart += length; }
            else if (OP_DELETE == code)
            {
                needCompacting = true;
                if (lessEq(binEndPos, plus(newPos, length))) return false;
                newPos = plus(newPos, length);
            }
            else throw std::logic_error("Unexpected CIGAR operation");
        }
        if (index.cigarEnd == cigarIterator) return false;            // the fragment gets completely soft-clipped
        const uint32_t *cigarBackwardsIterator = index.cigarEnd - 1;
        unsigned softClipEnd = 0;
        for (; cigarIterator != cigarBackwardsIterator; --cigarBackwardsIterator)
        {
            const unsigned length = opLength(*cigarBackwardsIterator), code = opCode(*cigarBackwardsIterator);
            if (OP_ALIGN == code) break;
            else if (OP_SOFT_CLIP == code) softClipEnd += length;
            else if (OP_INSERT == code) { needCompacting = true; softClipEnd += length; }
            else if (OP_DELETE == code) needCompacting = true;
            else throw std::logic_error("Unexpected CIGAR operation");
        }
        // the middle of the CIGAR as indexes: the buffer below may move
        std::vector<uint32_t> middle(cigarIterator, cigarBackwardsIterator + 1);
        if (needCompacting)
        {
            const size_t before = realignedCigars.size();
            if (softClipStart) realignedCigars.push_back(encode(softClipStart, OP_SOFT_CLIP));
            realignedCigars.insert(realignedCigars.end(), middle.begin(), middle.end());
            if (softClipEnd) realignedCigars.push_back(encode(softClipEnd, OP_SOFT_CLIP));
            index.cigarBegin = &realignedCigars.at(before); index.cigarEnd = &realignedCigars.back() + 1;
            index.pos = newPos;
        }
        // recompute editDistance and observed length
        unsigned short newEditDistance = 0;
        const unsigned char *basesIterator = fragment.bases + softClipStart;
        ReferencePosition newEndPos = index.pos;
        for (const uint32_t w : middle)
        {
            const unsigned length = opLength(w), code = opCode(w);
            if (OP_ALIGN == code) { newEditDistance += countMismatches(reference, basesIterator, newEndPos, length); newEndPos = plus(newEndPos, length); basesIterator += length; }
            else if (OP_INSERT == code) { newEditDistance += length; basesIterator += length; }
            else if (OP_DELETE == code) { newEditDistance += length; newEndPos = plus(newEndPos, length); }
            else throw std::logic_error("Unexpected CIGAR operation");
        }
        fragment.editDistance = newEditDistance;
        fragment.fStrandPosition = index.pos;
        fragment.observedLength = unsigned(minus(newEndPos, fragment.fStrandPosition));
        return true;
    }

    // GapRealigner::verifyGapsChoice (:505-651)
    GapChoice verifyGapsChoice(unsigned choice, const std::vector<RealignGap> &gaps, ReferencePosition newBeginPos, const RealignFragment &fragment)
    {
        GapChoice ret;
        int basesLeft = fragment.readLength;
        int leftClippedLeft = int(fragment.leftClipped());
        ReferencePosition lastGapEndPos = newBeginPos;
        ReferencePosition lastGapBeginPos;
        unsigned currentGapIndex = 0;
        for (const RealignGap &gap : gaps)
        {
            if (choice & (1u << currentGapIndex))
            {
                if (lessEq(gap.endPos(true), lastGapEndPos)) { ret.cost = -1U; return ret; }
                if (gap.pos < lastGapEndPos) { ret.cost = -1U; return ret; }
                if (gap.pos == lastGapBeginPos) { ret.cost = -1U; return ret; }
                const int mappedBases = std::min<int>(basesLeft - int(fragment.rightClipped()), int(minus(gap.pos, lastGapEndPos)));
                const unsigned length = mappedBases - std::min(mappedBases, leftClippedLeft);
                const unsigned mm = countMismatches(reference, fragment.bases + (fragment.readLength - basesLeft) + leftClippedLeft, plus(lastGapEndPos, leftClippedLeft), length);
                ret.mappedLength += length; ret.editDistance += mm; ret.mismatches += mm; ret.cost += mm * self.mismatchCost;
                basesLeft -= mappedBases;
                leftClippedLeft -= std::min(leftClippedLeft, mappedBases);
                unsigned clippedGapLength = 0;
                if (gap.isInsertion())
                {
                    clippedGapLength = std::min<int>(basesLeft - int(fragment.rightClipped()), gap.getLength());
                    basesLeft -= clippedGapLength;
                    leftClippedLeft -= std::min<int>(leftClippedLeft, gap.getLength());
                }
                else clippedGapLength = leftClippedLeft ? 0 : gap.getLength();
                ret.editDistance += clippedGapLength;
                ret.cost += clippedGapLength ? (self.gapOpenCost + (clippedGapLength - 1) * self.gapExtendCost) : 0;
                lastGapEndPos = gap.endPos(false);
                lastGapBeginPos = gap.pos;
                if (basesLeft == leftClippedLeft + int(fragment.rightClipped())) break;
                if (basesLeft < leftClippedLeft + int(fragment.rightClipped())) throw std::logic_error("Was not supposed to run into the clipping");
            }
            ++currentGapIndex;
        }
        if (basesLeft > leftClippedLeft + int(fragment.rightClipped()))
        {
            const unsigned length = basesLeft - std::min<unsigned>(basesLeft, leftClippedLeft) - fragment.rightClipped();
            const ReferencePosition firstUnclippedPos = plus(lastGapEndPos, leftClippedLeft);
            if (firstUnclippedPos.getPosition() > reference.at(firstUnclippedPos.getContigId()).forward.size()) { ret.cost = -1U; return ret; }
            const unsigned mm = countMismatches(reference, fragment.bases + (fragment.readLength - basesLeft) + leftClippedLeft, firstUnclippedPos, length);
            ret.mappedLength += length; ret.editDistance += mm; ret.mismatches += mm; ret.cost += mm * self.mismatchCost;
        }
        return ret;
    }

    // GapRealigner::applyChoice (:660-833)
    bool applyChoice(unsigned choice, const std::vector<RealignGap> &gaps, ReferencePosition binEndPos, ReferencePosition contigEndPos, RealignIndex &index, const RealignFragment &fragment)
    {
        ReferencePosition newBeginPos = index.pos;
        const size_t before = realignedCigars.size();
        int basesLeft = fragment.readLength;
        int leftClippedLeft = int(fragment.leftClipped());
        int leftClippedInsertionBases = 0;
        if (fragment.leftClipped()) realignedCigars.push_back(encode(fragment.leftClipped(), OP_SOFT_CLIP));
        ReferencePosition lastGapEndPos = newBeginPos;
        unsigned currentGapIndex = 0;
        unsigned lastOperation = 9;       // Cigar::UNKNOWN
        for (const RealignGap &gap : gaps)
        {
            if (choice & (1u << currentGapIndex))
            {
                const ReferencePosition gapClippedBeginPos = std::max(gap.pos, newBeginPos);
                if (!(gapClippedBeginPos < lastGapEndPos))
                {
                    const int mappedBases = std::min<int>(basesLeft - int(fragment.rightClipped()), int(minus(gapClippedBeginPos, lastGapEndPos)));
                    const unsigned softClippedMappedLength = mappedBases - std::min(mappedBases, leftClippedLeft);
                    if (softClippedMappedLength) realignedCigars.push_back(encode(softClippedMappedLength, OP_ALIGN));
                    basesLeft -= mappedBases;
                    leftClippedLeft -= std::min(mappedBases, leftClippedLeft);
                    if (gap.isInsertion())
                    {
                        const int clippedGapLength = std::min<int>(basesLeft - int(fragment.rightClipped()), int(minus(gap.endPos(true), gapClippedBeginPos)));
                        const int softClippedGapLength = clippedGapLength - std::min(clippedGapLength, leftClippedLeft);
                        if (softClippedGapLength)
                        {
                            if (OP_INSERT == lastOperation && !mappedBases) realignedCigars.back() = encode(opLength(realignedCigars.back()) + softClippedGapLength, OP_INSERT);
                            else { realignedCigars.push_back(encode(softClippedGapLength, OP_INSERT)); lastOperation = OP_INSERT; }
                        }
                        basesLeft -= clippedGapLength;
                        lastGapEndPos = gapClippedBeginPos;
                        leftClippedLeft -= std::min(clippedGapLength, leftClippedLeft);
                        leftClippedInsertionBases += clippedGapLength - softClippedGapLength;
                    }
                    else
                    {
                        const int clippedGapLength = int(minus(gap.endPos(true), gapClippedBeginPos));
                        if (!leftClippedLeft)
                        {
                            if (OP_DELETE == lastOperation && !mappedBases) realignedCigars.back() = encode(opLength(realignedCigars.back()) + clippedGapLength, OP_DELETE);
                            else { realignedCigars.push_back(encode(clippedGapLength, OP_DELETE)); lastOperation = OP_DELETE; }
                        }
                        else newBeginPos = plus(newBeginPos, clippedGapLength);
                        lastGapEndPos = gap.endPos(false);
                    }
                }
                else throw std::logic_error("Overlapping gaps are not allowed");
                if (basesLeft == leftClippedLeft + int(fragment.rightClipped())) break;
                if (basesLeft < leftClippedLeft + int(fragment.rightClipped())) throw std::logic_error("Was not supposed to run into the clipping");
            }
            ++currentGapIndex;
        }
        if (basesLeft > leftClippedLeft + int(fragment.rightClipped()))
        {
            const int basesToTheEndOfContig = int(minus(contigEndPos, lastGapEndPos)) - leftClippedLeft;
            const int mappedBases = std::min(basesToTheEndOfContig, basesLeft - leftClippedLeft - int(fragment.rightClipped()));
            if (mappedBases) realignedCigars.push_back(encode(mappedBases, OP_ALIGN));
            basesLeft -= leftClippedLeft + mappedBases;
            leftClippedLeft = 0;
        }
        if (basesLeft) realignedCigars.push_back(encode(basesLeft, OP_SOFT_CLIP));
        newBeginPos = plus(newBeginPos, fragment.leftClipped() - leftClippedInsertionBases);
        if (!(newBeginPos < binEndPos)) { realignedCigars.resize(before); return false; }
        index.pos = newBeginPos;
        index.cigarBegin = &realignedCigars.at(before); index.cigarEnd = &realignedCigars.back() + 1;
        return true;
    }

    // GapRealigner::findStartPos (:842-968)
    bool findStartPos(unsigned choice, const std::vector<RealignGap> &gaps, ReferencePosition binStartPos, ReferencePosition binEndPos, const RealignIndex &index, unsigned pivotGapIndex,
                      ReferencePosition pivotPos, ReferencePosition &ret)
    {
        ReferencePosition lastGapEndPos = unclippedPosition(index);
        long offset = minus(pivotPos, index.pos);
        for (const uint32_t *it = index.cigarBegin; index.cigarEnd != it; ++it)
        {
            if (pivotPos < lastGapEndPos) break;
            const unsigned length = opLength(*it), code = opCode(*it);
            if (OP_ALIGN == code) lastGapEndPos = plus(lastGapEndPos, length);
            else if (OP_INSERT == code) offset += length;
            else if (OP_DELETE == code)
            {
                lastGapEndPos = plus(lastGapEndPos, length);
                if (pivotPos < lastGapEndPos) return false;       // an existing deletion overlaps the pivot position
                offset -= length;
            }
            else if (OP_SOFT_CLIP == code)
            {
                if (index.cigarBegin == it) offset += length;
                lastGapEndPos = plus(lastGapEndPos, length);
            }
            else throw std::logic_error("Unexpected CIGAR operation");
        }
        if (0 > offset) return false;
        unsigned gapIndex = pivotGapIndex - 1;
        ReferencePosition overlapPos = pivotPos;
        unsigned basesLeft = unsigned(offset);
        for (size_t k = pivotGapIndex; k-- > 0;)
        {
            const RealignGap &gap = gaps[k];
            if (choice & (1u << gapIndex))
            {
                if (overlapPos < gap.endPos(false)) return false;                 // overlapping gaps are not allowed
                if (gap.isInsertion())
                {
                    const unsigned insertionBases = std::min(basesLeft, gap.getLength());
                    offset -= insertionBases; basesLeft -= insertionBases;
                    if (!basesLeft) break;
                }
                else { offset += gap.getLength(); overlapPos = gap.pos; }
            }
            --gapIndex;
        }
        // ReferencePosition arithmetic: the comparisons of the reference are on the encoded values
        if (pivotPos < plus(binStartPos, offset)) return false;
        if (!(plus(pivotPos, -offset) < binEndPos)) return false;
        ret = plus(pivotPos, -offset);
        return true;
    }
};

namespace
{
// getTotalGapsLength / calculateMismatchesPercent / GapRealigner::getAlignmentCost (:970-1040)
int calculateMismatchesPercent(unsigned mismatches, unsigned mappedLength) { return int(mismatches * 100 / mappedLength); }
}

unsigned GapRealigner::getAlignmentCost(const RealignFragment &fragment, const RealignIndex &index, unsigned &editDistance, int &mismatchesPercent) const
{
    unsigned gapsCount = 0, mappedLength = 0;
    unsigned short totalGapsLength = 0;
    for (const uint32_t *it = index.cigarBegin; it != index.cigarEnd; ++it)
    {
        const unsigned length = opLength(*it), code = opCode(*it);
        if (OP_ALIGN == code) mappedLength += length;
        else if (OP_INSERT == code || OP_DELETE == code) { totalGapsLength += length; ++gapsCount; }
        else if (OP_SOFT_CLIP != code) throw std::logic_error("Unexpected CIGAR operation");
    }
    editDistance = fragment.editDistance;
    const unsigned mismatches = fragment.editDistance - totalGapsLength;
    mismatchesPercent = calculateMismatchesPercent(mismatches, mappedLength);
    return mismatches * mismatchCost + gapsCount * gapOpenCost + gapExtendCost * (totalGapsLength - gapsCount);
}

// GapRealigner::realign (:1053-1268) without updatePairDetails, which needs the mate: the caller does it (realignPairDetails) when `changed` comes back true
void GapRealigner::realign(const RealignerGaps &realignerGaps, ReferencePosition binStartPos, ReferencePosition binEndPos, RealignIndex &index, RealignFragment &fragment,
                           std::vector<uint32_t> &realignedCigars, bool &changed) const
{
    changed = false;
    if (fragment.flags & 2) return;                     // unmapped
    if (realignedCigars.capacity() - realignedCigars.size() < 8192) throw std::logic_error("Realigned CIGAR buffer is out of capacity");
    const size_t bufferSizeBeforeRealignment = realignedCigars.size();
    Impl impl = { *this, reference, realignedCigars };
    std::vector<RealignGap> gaps;
    bool makesSenseToTryAgain = false;
    do
    {
        makesSenseToTryAgain = false;
        binEndPos = ReferencePosition(binEndPos.getContigId(), std::min<uint64_t>(binEndPos.getPosition(), reference.at(binEndPos.getContigId()).getLength()));
        const uint16_t DODGY = 0xffff;
        const ReferencePosition matePos = ReferencePosition::fromValue(fragment.mateFStrandPosition);
        if (fragment.editDistance &&
            (!(fragment.flags & 1) || (!(fragment.flags & 4) && lessEq(binStartPos, matePos) && matePos < binEndPos)) &&
            (realignDodgyFragments || DODGY != fragment.alignmentScore || DODGY != fragment.templateAlignmentScore) &&
            (index.pos.getPosition() >= beginClippedLength(index)))
        {
            index.pos = fragment.fStrandPosition;
            const RealignmentBounds bounds = extractRealignmentBounds(index);
            realignerGaps.findGaps(bounds.beginPos, bounds.endPos, gaps, MAX_GAPS_AT_A_TIME * 10);
            if (!realignGapsVigorously && MAX_GAPS_AT_A_TIME < gaps.size()) break;
            const OverlappingGapsFilter overlappingGapsFilter(gaps);
            unsigned bestEditDistance = 0;
            int originalMismatchesPercent = 0;
            unsigned bestCost = getAlignmentCost(fragment, index, bestEditDistance, originalMismatchesPercent);
            ReferencePosition bestStartPos = index.pos;
            unsigned bestChoice = 0, evaluatedSoFar = 0;
            const auto isBetterChoice = [&](const GapChoice &choice)
            {
                return choice.mappedLength && (choice.cost < bestCost || (choice.cost == bestCost && choice.editDistance < bestEditDistance)) &&
                       calculateMismatchesPercent(choice.mismatches, choice.mappedLength) <= originalMismatchesPercent;
            };
            for (unsigned choice = 0; (choice = overlappingGapsFilter.next(choice));)
            {
                if (((1u << MAX_GAPS_AT_A_TIME) - 1) < evaluatedSoFar++) break;
                unsigned pivotGapIndex = 0;
                for (const RealignGap &pivotGap : gaps)
                {
                    if (choice & (1u << pivotGapIndex))
                    {
                        ReferencePosition newStartPos;
                        if (lessEq(binStartPos, pivotGap.pos))
                        {
                            if (impl.findStartPos(choice, gaps, binStartPos, binEndPos, index, pivotGapIndex, pivotGap.pos, newStartPos))
                            {
                                const GapChoice thisChoice = impl.verifyGapsChoice(choice, gaps, newStartPos, fragment);
                                if (isBetterChoice(thisChoice)) { bestEditDistance = thisChoice.editDistance; bestChoice = choice; bestStartPos = newStartPos; bestCost = thisChoice.cost; }
                            }
                        }
                        if (impl.findStartPos(choice, gaps, binStartPos, binEndPos, index, pivotGapIndex + 1, pivotGap.endPos(false), newStartPos))
                        {
                            const GapChoice thisChoice = impl.verifyGapsChoice(choice, gaps, newStartPos, fragment);
                            if (isBetterChoice(thisChoice)) { bestEditDistance = thisChoice.editDistance; bestChoice = choice; bestStartPos = newStartPos; bestCost = thisChoice.cost; }
                        }
                    }
                    ++pivotGapIndex;
                }
            }
            if (bestChoice && bestStartPos < binEndPos)
            {
                RealignIndex tmp = index;
                tmp.pos = bestStartPos;
                const ReferencePosition contigEndPos(binEndPos.getContigId(), reference.at(binEndPos.getContigId()).forward.size());
                if (impl.applyChoice(bestChoice, gaps, binEndPos, contigEndPos, tmp, fragment))
                {
                    if (impl.compactCigar(binEndPos, tmp, fragment))
                    {
                        if (clipSemialigned) { impl.clipLeftSide(binEndPos, tmp, fragment); impl.clipRightSide(tmp, fragment); }
                        index = tmp;
                        changed = true;
                        makesSenseToTryAgain = realignGapsVigorously;
                    }
                }
            }
        }
    } while (makesSenseToTryAgain);
    // GapRealigner::compactRealignedCigarBuffer (:1270-1288)
    if (realignedCigars.size() != bufferSizeBeforeRealignment)
    {
        const size_t cigarLength = size_t(index.cigarEnd - index.cigarBegin), expectedBufferSize = bufferSizeBeforeRealignment + cigarLength;
        if (expectedBufferSize != realignedCigars.size())
        {
            const std::vector<uint32_t> kept(index.cigarBegin, index.cigarEnd);
            std::copy(kept.begin(), kept.end(), realignedCigars.begin() + bufferSizeBeforeRealignment);
            realignedCigars.resize(expectedBufferSize);
        }
        if (cigarLength) { index.cigarBegin = &realignedCigars[bufferSizeBeforeRealignment]; index.cigarEnd = index.cigarBegin + cigarLength; }
    }
}

} // namespace oracle
