// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Reads, quality tables, AlignerBase/Ungapped/Gapped/SimpleIndel aligners and FragmentBuilder.
#include "oracle.hpp"
#include <cmath>
#include <algorithm>
#include <stdexcept>
#include <cassert>
#include <numeric>

namespace oracle
{

std::string cigarToString(const uint32_t *b, const uint32_t *e)
{
    static const char ops[] = { 'M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X', '?' };
    std::string s;
    for (; b != e; ++b) { std::pair<unsigned, CigarOp> d = cigarDecode(*b); s += std::to_string(d.first); s.push_back(ops[d.second]); }
    return s;
}

// include/alignment/SeedId.hh:84-109
SeedId::SeedId(uint64_t tile, uint64_t barcode, uint64_t cluster, uint64_t seed, uint64_t reverse)
    : value(((tile & TILE_MASK) << TILE_SHIFT) | ((barcode & BARCODE_MASK) << BARCODE_SHIFT) | ((cluster & CLUSTER_MASK) << CLUSTER_SHIFT) |
            ((seed & SEED_MASK) << SEED_SHIFT) | (reverse & REVERSE_MASK))
{
    if ((TILE_MASK < tile) | (BARCODE_MASK < barcode) | (CLUSTER_MASK < cluster) | (SEED_MASK < seed) | (REVERSE_MASK < reverse))
        throw std::invalid_argument("SeedId: field overflow");
}

size_t genomeLength(const ContigList &c) { size_t r = 0; for (size_t i = 0; i < c.size(); ++i) r += c[i].getLength(); return r; }

// ---------------------------------------------------------------- Quality (lib/alignment/Quality.cpp:34-66)
double Quality::getLogMismatch(unsigned q) { const double mismatch = pow(10.0, (double)q / -10.0); return log(mismatch / 3.0); }
// (function-local statics initialised by a call: thread-safe in C++11.  Filling them lazily under `if (lookup.empty())` let the worker
// threads of oracle_select race on the first call of a process: a thread could read a table another one was still growing.)
static std::vector<double> makeLogMatchLookup()
{
    std::vector<double> lookup;
    const double nMismatch = pow(10.0, 1.0 / -10.0);
    lookup.push_back(log(1.0 - nMismatch));
    for (int i = 1; i < 100; ++i) { const double mismatch = pow(10.0, (double)i / -10.0); lookup.push_back(log(1.0 - mismatch)); }
    return lookup;
}
static std::vector<double> makeLogMismatchLookup()
{
    std::vector<double> lookup;
    lookup.push_back(log(1.0 - pow(10.0, 1.0 / -10.0)));
    for (unsigned q = 1; q < 100U; ++q) lookup.push_back(Quality::getLogMismatch(q));
    return lookup;
}
const std::vector<double> &Quality::logMatchLookup() { static const std::vector<double> lookup = makeLogMatchLookup(); return lookup; }
const std::vector<double> &Quality::logMismatchLookup() { static const std::vector<double> lookup = makeLogMismatchLookup(); return lookup; }
// include/alignment/Quality.hh:87-91 (genomeLength is passed through `unsigned`)
double Quality::restOfGenomeCorrection(unsigned genomeLength, unsigned readLength)
{
    return exp(log(2.0) + log((double)genomeLength) - (log(4.0) * (double)readLength));
}

// ---------------------------------------------------------------- Read / Cluster
// lib/alignment/Read.cpp:32-73; oligo/Nucleotides.hh:91-94 (isBclN)
void Read::decodeBcl(const uint8_t *b, const uint8_t *e, unsigned idx)
{
    static const char bases[] = { 'A', 'C', 'G', 'T' };
    index = idx;
    forwardSequence.clear(); reverseSequence.clear(); forwardQuality.clear(); reverseQuality.clear();
    endCyclesMasked = 0;
    for (const uint8_t *bcl = b; e > bcl; ++bcl)
    {
        if (*bcl & 0xfc)
        {
            forwardSequence.push_back(bases[*bcl & 3]);
            reverseSequence.push_back(bases[(~*bcl) & 3]);
            forwardQuality.push_back(char(*bcl >> 2));
            reverseQuality.push_back(char(*bcl >> 2));
        }
        else
        {
            forwardSequence.push_back('n'); reverseSequence.push_back('n');
            forwardQuality.push_back(2); reverseQuality.push_back(2);
        }
    }
    std::reverse(reverseSequence.begin(), reverseSequence.end());
    std::reverse(reverseQuality.begin(), reverseQuality.end());
}

// lib/alignment/Cluster.cpp:43-70 (barcodeLength == 0)
void Cluster::init(const std::vector<ReadMetadata> &readMetadata, const uint8_t *bclData, unsigned tile_, uint64_t id_, bool pf_)
{
    tile = tile_; id = id_; pf = pf_; bcl = bclData; nReads = 0;
    for (size_t i = 0; i < readMetadata.size(); ++i)
    {
        const ReadMetadata &rm = readMetadata[i];
        if (rm.length)
        {
            reads[rm.index].decodeBcl(bclData, bclData + rm.length, rm.index);
            bclData += rm.length;
            ++nReads;
        }
    }
}

// lib/alignment/Quality.cpp:71-105
void trimLowQualityEnd(Read &read, unsigned baseQualityCutoff)
{
    const unsigned MASK_READ_LENGTH_MIN = 35;
    if (read.getLength() < MASK_READ_LENGTH_MIN) return;
    const std::vector<char> &reverse = read.reverseQuality;
    int qscoreSum = 0, peakSum = 0;
    bool trimPosSet = false;
    size_t trimPos = 0;
    for (size_t it = 0; reverse.size() - MASK_READ_LENGTH_MIN != it; ++it)
    {
        qscoreSum += baseQualityCutoff - reverse[it];
        if (qscoreSum < 0) break;
        if (qscoreSum > peakSum) { peakSum = qscoreSum; trimPos = it; trimPosSet = true; }
    }
    if (trimPosSet) read.endCyclesMasked = unsigned(trimPos + 1);
}
void trimLowQualityEnds(Cluster &cluster, unsigned baseQualityCutoff)
{
    if (!baseQualityCutoff) return;
    for (unsigned r = 0; cluster.nReads > r; ++r) trimLowQualityEnd(cluster[r], baseQualityCutoff);
}

unsigned FragmentMetadata::getMappedLength() const
{
    unsigned ret = 0;
    for (unsigned i = 0; i < cigarLength; ++i) { std::pair<unsigned, CigarOp> d = cigarDecode((*cigarBuffer)[cigarOffset + i]); if (ALIGN == d.second) ret += d.first; }
    return ret;
}

// ---------------------------------------------------------------- sequencing adapters
// lib/alignment/matchSelector/SequencingAdapter.cpp:30-56: where each 5-mer of the adapter starts; a 5-mer that occurs twice is of no use
SequencingAdapter::SequencingAdapter(const SequencingAdapterMetadata &m)
    : metadata(m), kmerPositions(getMaxKmer<unsigned>(adapterMatchBasesMin) + 1, char(UNINITIALIZED_POSITION))
{
    if (metadata.sequence.size() >= 127) throw std::logic_error("Adapter sequence is too long");
    if (!(metadata.isUnbounded() || metadata.sequence.size() <= metadata.clipLength)) throw std::logic_error("Clip length cannot be shorter than the adapter sequence");
    const char *begin = metadata.sequence.data();
    KmerGenerator<unsigned short> kmerGenerator(begin, begin + metadata.sequence.size(), adapterMatchBasesMin);
    const char *position = begin;
    unsigned short kmer = 0;
    while (kmerGenerator.next(kmer, position))
    {
        char &pos = kmerPositions.at(kmer);
        if (UNINITIALIZED_POSITION == pos) pos = char(position - begin);
        else if (NON_UNIQUE_KMER_POSITION != pos) pos = NON_UNIQUE_KMER_POSITION;
    }
}

// SequencingAdapter.cpp:58-141.  sequenceBegin is where the caller's search began (not the read's first base): an adapter that starts before it
// counts as "before the sequence"
std::pair<long, long> SequencingAdapter::getMatchRange(const char *sequence, const long sequenceBegin, const long sequenceEnd, const long mismatchBase) const
{
    unsigned short kmer = 0;
    if (generateKmer(adapterMatchBasesMin, kmer, sequence + mismatchBase, sequence + sequenceEnd))
    {
        const char pos = kmerPositions[kmer];
        if (0 <= pos)
        {
            const unsigned mismatchBaseOffset = unsigned(mismatchBase - sequenceBegin);
            const unsigned adapterBasesBeforeSequence = mismatchBaseOffset < unsigned(pos) ? pos - mismatchBaseOffset : 0;
            if (!adapterBasesBeforeSequence || !metadata.isUnbounded())
            {
                const long testBase = mismatchBase - (pos - long(adapterBasesBeforeSequence));
                const unsigned testSequenceLength = unsigned(sequenceEnd - testBase);
                const unsigned adapterSequenceSize = unsigned(metadata.sequence.size());
                const unsigned leftClippedAdapterLength = adapterSequenceSize - adapterBasesBeforeSequence;
                const unsigned overlapLength = std::min<unsigned>(testSequenceLength, leftClippedAdapterLength);
                if (overlapLength < leftClippedAdapterLength && metadata.isUnbounded() && metadata.reverse)
                {
                    // unbounded adapter begins after the reverse sequence: ignored
                }
                else if (overlapLength >= adapterMatchBasesMin && !metadata.sequence.compare(adapterBasesBeforeSequence, overlapLength, sequence + testBase, overlapLength))
                {
                    if (metadata.reverse)
                        return metadata.isUnbounded() ?
                            std::make_pair(sequenceBegin, testBase + long(overlapLength)) :
                            std::make_pair(testBase - long(std::min<unsigned>(unsigned(testBase - sequenceBegin), metadata.clipLength - adapterSequenceSize)), testBase + long(overlapLength));
                    return metadata.isUnbounded() ?
                        std::make_pair(testBase, sequenceEnd) :
                        std::make_pair(testBase, testBase + long(std::min(overlapLength, metadata.clipLength)));
                }
            }
        }
    }
    return std::make_pair(mismatchBase, mismatchBase);
}

// FragmentSequencingAdapterClipper.cpp:76-97: at every base that does not match the reference, is this where an adapter shows?
static std::pair<long, long> findSequencingAdapter(const char *sequence, const long sequenceBegin, const long sequenceEnd, const std::vector<char> &reference, long referenceBegin,
                                                   const SequencingAdapter &adapter)
{
    for (long currentBase = sequenceBegin, currentReference = referenceBegin; sequenceEnd != currentBase; ++currentReference, ++currentBase)
    {
        if (!isMatch(sequence[currentBase], reference[currentReference]))
        {
            const std::pair<long, long> adapterMatchRange = adapter.getMatchRange(sequence, sequenceBegin, sequenceEnd, currentBase);
            if (adapterMatchRange.first != adapterMatchRange.second) return adapterMatchRange;
        }
    }
    return std::make_pair(sequenceBegin, sequenceBegin);
}

// FragmentSequencingAdapterClipper.cpp:103-150 (clipReference of :41-58 inlined): the first fragment of a strand that comes by decides where the
// strand's adapter lies, for every later fragment of that strand as well
void FragmentSequencingAdapterClipper::checkInitStrand(const FragmentMetadata &fragmentMetadata, const Contig &contig)
{
    const bool reverse = fragmentMetadata.reverse;
    Range &range = strandRange[reverse];
    if (range.initialized) return;
    const std::vector<char> &reference = contig.forward;
    const std::vector<char> &sequence = fragmentMetadata.getRead().getStrandSequence(reverse);
    long sequenceBegin = 0, sequenceEnd = long(sequence.size());
    const long referenceLeft = long(reference.size()) - fragmentMetadata.position;
    if (referenceLeft < sequenceEnd - sequenceBegin) sequenceEnd = sequenceBegin + referenceLeft;
    long newFragmentPos = fragmentMetadata.position;
    if (0 > fragmentMetadata.position) { sequenceBegin -= fragmentMetadata.position; newFragmentPos = 0; }
    range.begin = sequenceEnd;
    range.end = sequenceBegin;
    for (const SequencingAdapter &adapter : sequencingAdapters)
    {
        if (!adapter.isStrandCompatible(reverse)) continue;
        // (a read that lies wholly outside the contig: the reference's loop would run off its iterators; no candidate is made like that)
        if (range.end >= sequenceEnd) continue;
        const std::pair<long, long> adapterMatchRange = findSequencingAdapter(sequence.data(), range.end, sequenceEnd, reference, newFragmentPos + (range.end - sequenceBegin), adapter);
        if (adapterMatchRange.first != adapterMatchRange.second)
        {
            range.begin = std::min(adapterMatchRange.first, range.begin);
            range.end = std::max(adapterMatchRange.second, range.end);
        }
    }
    range.initialized = true;
    range.empty = sequenceBegin == range.end;
}

// Alignment.hh:91-146 over offsets; a reference index outside the contig compares as 'N'.  (The reference forms reference.begin() + contigPosition
// without looking: for a fragment other than the strand's first one that hangs over the contig's start it reads what lies in front of the vector --
// undefined there, defined here, the same way in the device code.)
static char referenceBaseOrN(const std::vector<char> &reference, long i) { return (i < 0 || i >= long(reference.size())) ? 'N' : reference[i]; }
static unsigned countAdapterMatches(const char *sequence, long sequenceBegin, long sequenceEnd, const std::vector<char> &reference, long referenceBegin, long referenceEnd, bool matches)
{
    unsigned ret = 0;
    for (; sequenceEnd != sequenceBegin && referenceEnd != referenceBegin; ++sequenceBegin, ++referenceBegin)
        ret += matches == isMatch(sequence[sequenceBegin], referenceBaseOrN(reference, referenceBegin));
    return ret;
}
static unsigned percentMismatches(const char *sequence, long sequenceBegin, long sequenceEnd, const std::vector<char> &reference, long referenceBegin, long referenceEnd)
{
    const unsigned overlapLength = unsigned(std::min(sequenceEnd - sequenceBegin, referenceEnd - referenceBegin));
    return countAdapterMatches(sequence, sequenceBegin, sequenceEnd, reference, referenceBegin, referenceEnd, false) * 100 / overlapLength;
}

// FragmentSequencingAdapterClipper.cpp:152-222
bool FragmentSequencingAdapterClipper::decideWhichSideToClip(const Contig &contig, const long contigPosition, const char *sequence, const long sequenceLengthL, const Range &range,
                                                             bool &clipBackwards)
{
    const unsigned backwardsClipped = unsigned(range.begin);
    const unsigned forwardsClipped = unsigned(sequenceLengthL - range.end);
    clipBackwards = backwardsClipped < forwardsClipped;
    const unsigned sequenceLength = unsigned(sequenceLengthL);
    const std::vector<char> &reference = contig.forward;
    // abs(unsigned - unsigned): the only abs in scope in the reference's translation unit is ::abs(int) of <stdlib.h> (with more than one it would not
    // compile: the call is ambiguous for an unsigned argument), so the difference wraps to int and this is |backwards - forwards| < 9
    if (backwardsClipped && forwardsClipped && std::abs(int(backwardsClipped - forwardsClipped)) < 9)
    {
        if (contigPosition >= 0 && reference.size() >= unsigned(contigPosition + sequenceLength))
        {
            const long referenceBegin = contigPosition, referenceEnd = referenceBegin + sequenceLength;
            const unsigned backwardsMatches = countAdapterMatches(sequence, 0, range.begin, reference, referenceBegin, referenceBegin + backwardsClipped, true);
            const unsigned forwardsMatches = countAdapterMatches(sequence, range.end, sequenceLengthL, reference, referenceEnd - forwardsClipped, referenceEnd, true);
            clipBackwards = (backwardsMatches < forwardsMatches || (backwardsMatches == forwardsMatches && backwardsClipped < forwardsClipped));
        }
    }
    else if (!backwardsClipped || !forwardsClipped)
    {
        const long referenceBegin = contigPosition, referenceEnd = referenceBegin + sequenceLength;
        if (clipBackwards && !backwardsClipped)
        {
            const unsigned basesClipped = unsigned(range.end);
            return percentMismatches(sequence, 0, range.end, reference, referenceBegin, referenceBegin + basesClipped) > TOO_GOOD_READ_MISMATCH_PERCENT;
        }
        else if (!clipBackwards && !forwardsClipped)
        {
            const unsigned basesClipped = unsigned(sequenceLengthL - range.begin);
            return percentMismatches(sequence, range.begin, sequenceLengthL, reference, referenceEnd - basesClipped, referenceEnd) > TOO_GOOD_READ_MISMATCH_PERCENT;
        }
    }
    return true;
}

// FragmentSequencingAdapterClipper.cpp:230-278
void FragmentSequencingAdapterClipper::clip(const Contig &contig, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd) const
{
    const Range &range = strandRange[fragment.reverse];
    if (!range.initialized) throw std::logic_error("checkInitStrand has not been called");
    if (range.empty) return;
    const long sequenceLength = sequenceEnd - sequenceBegin;
    if (range.begin < 0 || range.begin > sequenceLength || range.end < 0 || range.end > sequenceLength) throw std::logic_error("adapter range is outside the sequence");
    bool clipBackwards = false;
    if (decideWhichSideToClip(contig, fragment.position, sequenceBegin, sequenceLength, range, clipBackwards))
    {
        if (clipBackwards)
        {
            fragment.incrementClipLeft((unsigned short)(range.end));
            sequenceBegin += range.end;
        }
        else
        {
            fragment.incrementClipRight((unsigned short)(sequenceLength - range.begin));
            sequenceEnd = sequenceBegin + range.begin;
        }
    }
}

// ---------------------------------------------------------------- AlignerBase
// lib/alignment/fragmentBuilder/AlignerBase.cpp:50-82
void AlignerBase::clipReference(long referenceSize, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd)
{
    const long referenceLeft = referenceSize - fragment.position;
    if (referenceLeft >= 0)
    {
        if (referenceLeft < sequenceEnd - sequenceBegin) sequenceEnd = sequenceBegin + referenceLeft;
        if (0 > fragment.position) { sequenceBegin -= fragment.position; fragment.position = 0L; }
        sequenceEnd = std::max(sequenceEnd, sequenceBegin);
    }
    else
    {
        fragment.position += referenceLeft - 1;
        sequenceBegin += referenceLeft - 1;
        --sequenceBegin;
        sequenceEnd = sequenceBegin;
    }
}

// AlignerBase.cpp:89-119
void AlignerBase::clipReadMasking(const Read &read, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd)
{
    const char *maskedBegin, *maskedEnd;
    if (fragment.reverse)
    {
        maskedBegin = read.reverseSequence.data() + read.getEndCyclesMasked();
        maskedEnd = read.reverseSequence.data() + read.reverseSequence.size() - read.getBeginCyclesMasked();
    }
    else
    {
        maskedBegin = read.forwardSequence.data() + read.getBeginCyclesMasked();
        maskedEnd = read.forwardSequence.data() + read.forwardSequence.size() - read.getEndCyclesMasked();
    }
    if (maskedBegin > sequenceBegin) { fragment.incrementClipLeft((unsigned short)(maskedBegin - sequenceBegin)); sequenceBegin = maskedBegin; }
    if (maskedEnd < sequenceEnd) { fragment.incrementClipRight((unsigned short)(sequenceEnd - maskedEnd)); sequenceEnd = maskedEnd; }
}

// AlignerBase.cpp:121-227
unsigned AlignerBase::updateFragmentCigar(const std::vector<ReadMetadata> &reads, const std::vector<char> &reference, FragmentMetadata &f,
                                          long strandPosition, const Cigar &cigarBuffer, unsigned cigarOffset) const
{
    const Read &read = f.getRead();
    const bool reverse = f.reverse;
    const std::vector<char> &sequence = read.getStrandSequence(reverse);
    const std::vector<char> &quality = read.getStrandQuality(reverse);
    if (reference.empty()) throw std::logic_error("Reference contig was not loaded");
    if (0 > strandPosition) throw std::logic_error("position must be positive for CIGAR update");
    const char *currentReference = reference.data() + strandPosition;
    const unsigned firstCycle = reads[f.readIndex].firstCycle;
    const unsigned lastCycle = reads[f.readIndex].lastCycle();
    f.cigarBuffer = &cigarBuffer;
    f.cigarOffset = cigarOffset;
    f.cigarLength = unsigned(cigarBuffer.size()) - f.cigarOffset;
    unsigned currentBase = 0, matchCount = 0;
    for (unsigned i = 0; f.cigarLength > i; ++i)
    {
        const std::pair<unsigned, CigarOp> cigar = cigarDecode(cigarBuffer[f.cigarOffset + i]);
        const unsigned length = cigar.first;
        const CigarOp opCode = cigar.second;
        if (opCode == ALIGN)
        {
            unsigned matchesInARow = 0;
            for (unsigned j = 0; length > j; ++j)
            {
                if (isMatch(sequence[currentBase], *currentReference))
                {
                    ++matchCount; ++matchesInARow;
                    f.logProbability += Quality::getLogMatch((unsigned char)quality[currentBase]);
                }
                else
                {
                    f.matchesInARow = std::max(f.matchesInARow, matchesInARow);
                    matchesInARow = 0;
                    f.addMismatchCycle(reverse ? lastCycle - currentBase : firstCycle + currentBase);
                    f.logProbability += Quality::getLogMismatchFast((unsigned char)quality[currentBase]);
                    f.smithWatermanScore += normalizedMismatchScore;
                }
                if (sequence[currentBase] != *currentReference) ++f.editDistance;
                ++currentReference; ++currentBase;
            }
            f.matchesInARow = std::max(f.matchesInARow, matchesInARow);
        }
        else if (opCode == INSERT)
        {
            currentBase += length; f.editDistance += length; ++f.gapCount;
            f.smithWatermanScore += normalizedGapOpenScore + std::min(normalizedMaxGapExtendScore, (length - 1) * normalizedGapExtendScore);
        }
        else if (opCode == DELETE)
        {
            currentReference += length; f.editDistance += length; ++f.gapCount;
            f.smithWatermanScore += normalizedGapOpenScore + std::min(normalizedMaxGapExtendScore, (length - 1) * normalizedGapExtendScore);
        }
        else if (opCode == SOFT_CLIP)
        {
            double lp = f.logProbability;
            for (unsigned j = 0; j < length; ++j) lp = lp + Quality::getLogMatch((unsigned char)quality[currentBase + j]);
            f.logProbability = lp;
            currentBase += length;
        }
        else throw std::logic_error("Unexpected Cigar OpCode");
    }
    f.observedLength = unsigned(currentReference - reference.data() - strandPosition);
    f.position = strandPosition;
    if (currentBase != sequence.size()) throw std::logic_error("Unexpected discrepancy between cigar and sequence");
    return matchCount;
}

// lib/alignment/fragmentBuilder/UngappedAligner.cpp:39-92
unsigned UngappedAligner::alignUngapped(FragmentMetadata &f, Cigar &cigarBuffer, const std::vector<ReadMetadata> &reads, const FragmentSequencingAdapterClipper &adapterClipper,
                                        const Contig &contig) const
{
    const unsigned cigarOffset = unsigned(cigarBuffer.size());
    f.resetAlignment(cigarBuffer);
    f.resetClipping();
    const Read &read = f.getRead();
    const std::vector<char> &sequence = read.getStrandSequence(f.reverse);
    const std::vector<char> &reference = contig.forward;
    const char *sequenceBegin = sequence.data();
    const char *sequenceEnd = sequence.data() + sequence.size();
    adapterClipper.clip(contig, f, sequenceBegin, sequenceEnd);
    clipReadMasking(read, f, sequenceBegin, sequenceEnd);
    clipReference(long(reference.size()), f, sequenceBegin, sequenceEnd);
    const unsigned firstMappedBaseOffset = unsigned(sequenceBegin - sequence.data());
    if (firstMappedBaseOffset) cigarBuffer.push_back(cigarEncode(firstMappedBaseOffset, SOFT_CLIP));
    const unsigned mappedBases = unsigned(sequenceEnd - sequenceBegin);
    if (mappedBases) cigarBuffer.push_back(cigarEncode(mappedBases, ALIGN));
    const unsigned clipEndBases = unsigned(sequence.data() + sequence.size() - sequenceEnd);
    if (clipEndBases) cigarBuffer.push_back(cigarEncode(clipEndBases, SOFT_CLIP));
    const unsigned ret = updateFragmentCigar(reads, reference, f, f.position, cigarBuffer, cigarOffset);
    if (!ret) f.setUnaligned();
    return ret;
}

// lib/alignment/fragmentBuilder/GappedAligner.cpp:51-82
static std::pair<unsigned, unsigned> getFlanks(long strandPosition, unsigned readLength, unsigned long referenceSize, unsigned widestGapSize)
{
    if (strandPosition >= widestGapSize / 2)
    {
        if (strandPosition + readLength + (widestGapSize - widestGapSize / 2) < long(referenceSize))
        {
            const unsigned left = widestGapSize / 2;
            return std::make_pair(left, widestGapSize - left - 1);
        }
        const unsigned right = unsigned(referenceSize - readLength - strandPosition);
        return std::make_pair(widestGapSize - right - 1, right);
    }
    const unsigned left = unsigned(strandPosition);
    return std::make_pair(left, widestGapSize - left - 1);
}

// GappedAligner.cpp:167-249
unsigned GappedAligner::alignGapped(FragmentMetadata &f, Cigar &cigarBuffer, const std::vector<ReadMetadata> &reads, const FragmentSequencingAdapterClipper &adapterClipper,
                                    const Contig &contig) const
{
    const unsigned cigarOffset = unsigned(cigarBuffer.size());
    f.resetAlignment(cigarBuffer);
    f.resetClipping();
    const Read &read = f.getRead();
    const std::vector<char> &sequence = read.getStrandSequence(f.reverse);
    const std::vector<char> &reference = contig.forward;
    const char *sequenceBegin = sequence.data();
    const char *sequenceEnd = sequence.data() + sequence.size();
    adapterClipper.clip(contig, f, sequenceBegin, sequenceEnd);
    clipReadMasking(read, f, sequenceBegin, sequenceEnd);
    clipReference(long(reference.size()), f, sequenceBegin, sequenceEnd);
    const unsigned firstMappedBaseOffset = unsigned(sequenceBegin - sequence.data());
    if (firstMappedBaseOffset) cigarBuffer.push_back(cigarEncode(firstMappedBaseOffset, SOFT_CLIP));
    const unsigned sequenceLength = unsigned(sequenceEnd - sequenceBegin);
    long strandPosition = f.position;
    if (long(reference.size()) < long(sequenceLength) + strandPosition + long(BandedSmithWaterman::WIDEST_GAP_SIZE)) return 0;
    const std::pair<unsigned, unsigned> flanks = getFlanks(strandPosition, sequenceLength, reference.size(), BandedSmithWaterman::WIDEST_GAP_SIZE);
    const char *databaseBegin = reference.data() + strandPosition - flanks.first;
    const char *databaseEnd = databaseBegin + flanks.first + sequenceLength + flanks.second;
    strandPosition += bsw.align(sequenceBegin, sequenceEnd, databaseBegin, databaseEnd, cigarBuffer);
    const unsigned clipEndBases = unsigned(sequence.data() + sequence.size() - sequenceEnd);
    if (clipEndBases) cigarBuffer.push_back(cigarEncode(clipEndBases, SOFT_CLIP));
    strandPosition -= flanks.first;
    return updateFragmentCigar(reads, reference, f, strandPosition, cigarBuffer, cigarOffset);
}

// include/alignment/Alignment.hh:115-157
static unsigned countMismatches(const char *seq, const char *refBegin, const char *refEnd, unsigned length)
{
    unsigned ret = 0;
    const char *seqEnd = seq + length;
    for (; seqEnd != seq && refEnd != refBegin; ++seq, ++refBegin) ret += !isMatch(*seq, *refBegin);
    return ret;
}

// lib/alignment/fragmentBuilder/SimpleIndelAligner.cpp:50-229
void SimpleIndelAligner::alignSimpleDeletion(Cigar &cigarBuffer, FragmentMetadata &headAlignment, const unsigned headSeedOffset,
                                             FragmentMetadata &tailAlignment, const unsigned tailSeedOffset, const unsigned tailSeedLength,
                                             const ContigList &contigList, const std::vector<ReadMetadata> &reads) const
{
    if (headSeedOffset < headAlignment.getBeginClippedLength()) return;
    if (tailAlignment.getBeginClippedLength() + tailAlignment.getObservedLength() < tailSeedOffset + tailSeedLength) return;
    const unsigned tailOffset = headSeedOffset;
    const Read &read = headAlignment.getRead();
    const bool reverse = headAlignment.reverse;
    const char *sequenceBegin = read.getStrandSequence(reverse).data();
    const std::vector<char> &reference = contigList[headAlignment.contigId].forward;
    const char *refBegin = reference.data();
    const char *refEnd = reference.data() + reference.size();

    const char *tailIterator = sequenceBegin + tailOffset;
    unsigned tailLength = unsigned(tailAlignment.getBeginClippedLength() + tailAlignment.getObservedLength() - tailOffset);
    const unsigned tailMismatches = countMismatches(tailIterator, refBegin + headAlignment.getUnclippedPosition() + tailOffset, refEnd, tailLength);
    if (!tailMismatches) return;
    const long deletionLengthL = tailAlignment.getUnclippedPosition() - headAlignment.getUnclippedPosition();
    if (deletionLengthL < 0 || deletionLengthL > long(0xffffffffu)) throw std::range_error("bad numeric_cast"); // boost::numeric_cast<unsigned>
    const unsigned deletionLength = unsigned(deletionLengthL);

    unsigned rightRealignedMismatches = countMismatches(tailIterator, refBegin + tailAlignment.getUnclippedPosition() + tailOffset, refEnd, tailLength);
    unsigned leftRealignedMismatches = 0;
    unsigned leftFlankMismatches = countMismatches(tailIterator - std::min(GAP_FLANK_BASES, tailOffset),
                                                   refBegin + headAlignment.getUnclippedPosition() + tailOffset - std::min(32U, tailOffset), refEnd,
                                                   std::min(GAP_FLANK_BASES, tailOffset));
    unsigned rightFlankMismatches = countMismatches(tailIterator, refBegin + tailAlignment.getUnclippedPosition() + tailOffset, refEnd,
                                                    std::min(GAP_FLANK_BASES, tailLength));
    const char *referenceIterator = refBegin + headAlignment.getUnclippedPosition() + tailOffset;
    unsigned bestMismatches = tailMismatches, bestLeftFlankMismatches = leftFlankMismatches, bestRightFlankMismatches = rightFlankMismatches;
    unsigned bestOffset = -1U;
    for (unsigned deletionOffset = tailOffset; bestMismatches && deletionOffset <= tailSeedOffset;
         ++deletionOffset, ++tailIterator, ++referenceIterator, --tailLength)
    {
        const unsigned thisOffsetMismatches = leftRealignedMismatches + rightRealignedMismatches;
        if (bestMismatches > thisOffsetMismatches)
        {
            bestOffset = deletionOffset; bestMismatches = thisOffsetMismatches;
            bestLeftFlankMismatches = leftFlankMismatches; bestRightFlankMismatches = rightFlankMismatches;
        }
        const bool newLeftMismatch = !isMatch(*tailIterator, *referenceIterator);
        leftRealignedMismatches += newLeftMismatch;
        leftFlankMismatches += newLeftMismatch;
        if (deletionOffset >= GAP_FLANK_BASES)
            leftFlankMismatches -= !isMatch(*(tailIterator - GAP_FLANK_BASES), *(referenceIterator - GAP_FLANK_BASES));
        const bool disappearingRightMismatch = !isMatch(*tailIterator, *(referenceIterator + deletionLength));
        rightRealignedMismatches -= disappearingRightMismatch;
        rightFlankMismatches -= disappearingRightMismatch;
        if (tailLength > GAP_FLANK_BASES)
            rightFlankMismatches += !isMatch(*(tailIterator + GAP_FLANK_BASES), *(referenceIterator + deletionLength + GAP_FLANK_BASES));
    }
    if (bestLeftFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && bestRightFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && -1U != bestOffset)
    {
        const long clippingPositionOffset = headAlignment.getBeginClippedLength();
        const unsigned leftMapped = unsigned(bestOffset - clippingPositionOffset);
        const unsigned headMismatches = countMismatches(sequenceBegin + clippingPositionOffset, refBegin + headAlignment.position, refEnd, leftMapped);
        const unsigned newMismatches = headMismatches + bestMismatches;
        const unsigned sws = normalizedMismatchScore * newMismatches + normalizedGapOpenScore +
            std::min(normalizedMaxGapExtendScore, (deletionLength - 1) * normalizedGapExtendScore);
        if (headAlignment.smithWatermanScore > sws || (headAlignment.smithWatermanScore == sws && headAlignment.getMismatchCount() > newMismatches))
        {
            const unsigned cigarOffset = unsigned(cigarBuffer.size());
            if (clippingPositionOffset) cigarBuffer.push_back(cigarEncode(unsigned(clippingPositionOffset), SOFT_CLIP));
            if (leftMapped)
            {
                cigarBuffer.push_back(cigarEncode(leftMapped, ALIGN));
                cigarBuffer.push_back(cigarEncode(deletionLength, DELETE));
            }
            else headAlignment.position += deletionLength;
            const unsigned rightMapped = unsigned(headAlignment.getObservedLength() + headAlignment.getEndClippedLength() - leftMapped - tailAlignment.getEndClippedLength());
            if (rightMapped) cigarBuffer.push_back(cigarEncode(rightMapped, ALIGN));
            const unsigned clipEndBases = unsigned(tailAlignment.getEndClippedLength());
            if (clipEndBases) cigarBuffer.push_back(cigarEncode(clipEndBases, SOFT_CLIP));
            headAlignment.resetAlignment(cigarBuffer);
            headAlignment.rightClipped() = tailAlignment.rightClipped();
            if (!updateFragmentCigar(reads, reference, headAlignment, headAlignment.position + clippingPositionOffset, cigarBuffer, cigarOffset))
                throw std::logic_error("The alignment can't have no matches here");
        }
    }
}

// SimpleIndelAligner.cpp:241-438
void SimpleIndelAligner::alignSimpleInsertion(Cigar &cigarBuffer, FragmentMetadata &headAlignment, const unsigned headSeedOffset, const unsigned headSeedLength,
                                              FragmentMetadata &tailAlignment, const unsigned tailSeedOffset, const unsigned tailSeedLength,
                                              const ContigList &contigList, const std::vector<ReadMetadata> &reads) const
{
    if (headSeedOffset < headAlignment.getBeginClippedLength()) return;
    if (tailAlignment.getBeginClippedLength() + tailAlignment.getObservedLength() < tailSeedOffset + tailSeedLength) return;
    const unsigned tailOffset = headSeedOffset + headSeedLength;
    const unsigned observedEnd = unsigned(tailAlignment.getBeginClippedLength() + tailAlignment.getObservedLength());
    const long insertionLengthL = headAlignment.getUnclippedPosition() - tailAlignment.getUnclippedPosition();
    if (insertionLengthL < 0 || insertionLengthL > long(0xffffffffu)) throw std::range_error("bad numeric_cast");
    const unsigned insertionLength = unsigned(insertionLengthL);
    if (tailSeedOffset - headSeedOffset < insertionLength + headSeedLength) return;

    const Read &read = headAlignment.getRead();
    const bool reverse = headAlignment.reverse;
    const char *sequenceBegin = read.getStrandSequence(reverse).data();
    const std::vector<char> &reference = contigList[headAlignment.contigId].forward;
    const char *refBegin = reference.data();
    const char *refEnd = reference.data() + reference.size();

    const char *tailIterator = sequenceBegin + tailOffset + insertionLength;
    unsigned tailLength = observedEnd - tailOffset - insertionLength;
    const unsigned tailMismatches = countMismatches(tailIterator, refBegin + headAlignment.getUnclippedPosition() + tailOffset, refEnd, tailLength);
    unsigned leftFlankMismatches = countMismatches(tailIterator - insertionLength - GAP_FLANK_BASES,
                                                   refBegin + headAlignment.getUnclippedPosition() + tailOffset - GAP_FLANK_BASES, refEnd, GAP_FLANK_BASES);
    unsigned rightFlankMismatches = countMismatches(tailIterator, refBegin + headAlignment.getUnclippedPosition() + tailOffset, refEnd,
                                                    std::min(GAP_FLANK_BASES, tailLength));
    unsigned rightRealignedMismatches = tailMismatches;
    unsigned leftRealignedMismatches = 0;
    const char *referenceIterator = refBegin + headAlignment.getUnclippedPosition() + tailOffset;
    unsigned bestMismatches = tailMismatches, bestOffset = tailOffset;
    unsigned bestLeftFlankMismatches = leftFlankMismatches, bestRightFlankMismatches = rightFlankMismatches;
    for (unsigned insertionOffset = tailOffset; bestMismatches && insertionOffset <= tailSeedOffset - insertionLength;
         ++insertionOffset, ++tailIterator, ++referenceIterator, --tailLength)
    {
        const unsigned thisOffsetMismatches = leftRealignedMismatches + rightRealignedMismatches;
        if (bestMismatches > thisOffsetMismatches)
        {
            bestOffset = insertionOffset; bestMismatches = thisOffsetMismatches;
            bestLeftFlankMismatches = leftFlankMismatches; bestRightFlankMismatches = rightFlankMismatches;
        }
        const bool newLeftMismatch = !isMatch(*(tailIterator - insertionLength), *referenceIterator);
        leftRealignedMismatches += newLeftMismatch;
        leftFlankMismatches += newLeftMismatch;
        if (insertionOffset >= GAP_FLANK_BASES)
            leftFlankMismatches -= !isMatch(*(tailIterator - insertionLength - GAP_FLANK_BASES), *(referenceIterator - GAP_FLANK_BASES));
        const bool disappearingRightMismatch = !isMatch(*tailIterator, *referenceIterator);
        rightRealignedMismatches -= disappearingRightMismatch;
        rightFlankMismatches -= disappearingRightMismatch;
        if (tailLength > GAP_FLANK_BASES)
            rightFlankMismatches += !isMatch(*(tailIterator + GAP_FLANK_BASES), *(referenceIterator + GAP_FLANK_BASES));
    }
    const long clippingPositionOffset = headAlignment.getBeginClippedLength();
    const unsigned leftMapped = unsigned(bestOffset - clippingPositionOffset);
    if (!leftMapped) throw std::logic_error("Simple insertions are not allowed to be placed at the very beginning of the read");
    const unsigned headMismatches = countMismatches(sequenceBegin + clippingPositionOffset, refBegin + headAlignment.position, refEnd, leftMapped);
    const unsigned newMismatches = headMismatches + bestMismatches;
    const unsigned sws = normalizedMismatchScore * newMismatches + normalizedGapOpenScore +
        std::min(normalizedMaxGapExtendScore, (insertionLength - 1) * normalizedGapExtendScore);
    if (bestLeftFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && bestRightFlankMismatches <= GAP_FLANK_MISMATCHES_MAX)
    {
        if (tailAlignment.smithWatermanScore > sws || (tailAlignment.smithWatermanScore == sws && tailAlignment.getMismatchCount() > newMismatches))
        {
            const unsigned cigarOffset = unsigned(cigarBuffer.size());
            if (clippingPositionOffset) cigarBuffer.push_back(cigarEncode(unsigned(clippingPositionOffset), SOFT_CLIP));
            cigarBuffer.push_back(cigarEncode(leftMapped, ALIGN));
            cigarBuffer.push_back(cigarEncode(insertionLength, INSERT));
            const unsigned rightMapped = unsigned(headAlignment.getObservedLength() + headAlignment.getEndClippedLength() - leftMapped - tailAlignment.getEndClippedLength() - insertionLength);
            if (!rightMapped) throw std::logic_error("Simple insertions are not allowed to be placed at the very end of the read");
            cigarBuffer.push_back(cigarEncode(rightMapped, ALIGN));
            const unsigned clipEndBases = unsigned(tailAlignment.getEndClippedLength());
            if (clipEndBases) cigarBuffer.push_back(cigarEncode(clipEndBases, SOFT_CLIP));
            tailAlignment.resetAlignment(cigarBuffer);
            tailAlignment.leftClipped() = headAlignment.leftClipped();
            if (!updateFragmentCigar(reads, reference, tailAlignment, headAlignment.position, cigarBuffer, cigarOffset))
                throw std::logic_error("The alignment can't have no matches here");
        }
    }
}

// SimpleIndelAligner.cpp:443-449
static bool orderByUnclippedPosition(const FragmentMetadata &left, const FragmentMetadata &right)
{
    return left.contigId < right.contigId || (left.contigId == right.contigId && left.getUnclippedPosition() < right.getUnclippedPosition());
}

// SimpleIndelAligner.cpp:460-518
void SimpleIndelAligner::alignSimpleIndels(Cigar &cigarBuffer, const ContigList &contigList, const std::vector<ReadMetadata> &reads,
                                           const std::vector<SeedMetadata> &seedMetadataList, FragmentMetadataList &fragmentList) const
{
    if (fragmentList.size() < 2) return;
    std::sort(fragmentList.begin(), fragmentList.end(), orderByUnclippedPosition);
    FragmentMetadataList::iterator head = fragmentList.begin();
    for (FragmentMetadataList::iterator tail = head + 1; fragmentList.end() != tail; ++tail, ++head)
    {
        if (head->contigId == tail->contigId && head->reverse == tail->reverse)
        {
            const SeedMetadata &headSeed = seedMetadataList.at(head->firstSeedIndex);
            const SeedMetadata &tailSeed = seedMetadataList.at(tail->firstSeedIndex);
            const long distance = tail->getUnclippedPosition() - head->getUnclippedPosition();
            if (!distance) throw std::logic_error("distance must be non-zero for gap introduction");
            if (std::abs(distance) < long(semialignedGapLimit))
            {
                const long headSeedOffset = head->reverse ? long(head->getReadLength()) - headSeed.offset - headSeed.length : long(headSeed.offset);
                const long tailSeedOffset = head->reverse ? long(head->getReadLength()) - tailSeed.offset - tailSeed.length : long(tailSeed.offset);
                const long expectedSeedDistance = tailSeedOffset - headSeedOffset;
                if (0 < expectedSeedDistance)
                    alignSimpleDeletion(cigarBuffer, *head, unsigned(headSeedOffset), *tail, unsigned(tailSeedOffset), tailSeed.length, contigList, reads);
                else
                    alignSimpleInsertion(cigarBuffer, *tail, unsigned(tailSeedOffset), tailSeed.length, *head, unsigned(headSeedOffset), headSeed.length, contigList, reads);
            }
        }
    }
}

// ---------------------------------------------------------------- FragmentBuilder (lib/alignment/FragmentBuilder.cpp)
static unsigned maxSeedsPerRead(const Params &p)
{
    unsigned c[2] = { 0, 0 };
    for (size_t i = 0; i < p.seeds.size(); ++i) ++c[p.seeds[i].readIndex];
    return std::max(c[0], c[1]);
}

FragmentBuilder::FragmentBuilder(const Params &p)
    : repeatThreshold(p.repeatThreshold), semialignedGapLimit(p.semialignedGapLimit), gappedMismatchesMax(p.gappedMismatchesMax),
      seedMatchCounts(maxSeedsPerRead(p) * 2), repeatSeedsCount(0), fragments(2),
      ungappedAligner(p.gapMatchScore, p.gapMismatchScore, p.gapOpenScore, p.gapExtendScore, p.minGapExtendScore),
      gappedAligner(int(p.clusterLength()), p.gapMatchScore, p.gapMismatchScore, p.gapOpenScore, p.gapExtendScore, p.minGapExtendScore),
      simpleIndelAligner(p.gapMatchScore, p.gapMismatchScore, p.gapOpenScore, p.gapExtendScore, p.minGapExtendScore, p.semialignedGapLimit)
{
    for (const SequencingAdapterMetadata &m : p.adapters) sequencingAdapters.push_back(SequencingAdapter(m));
    // the reference reserves the cigar buffer so that it never reallocates (pointers into it stay valid): FragmentBuilder.cpp:56-58
    cigarBuffer.reserve(1 << 16);
}

void FragmentBuilder::clear()
{
    fragments[0].clear(); fragments[1].clear(); cigarBuffer.clear();
    std::fill(seedMatchCounts.begin(), seedMatchCounts.end(), 0);
    repeatSeedsCount = 0;
}

// FragmentBuilder.cpp:326-343
static long getReadPosition(const std::vector<ReadMetadata> &reads, const SeedMetadata &seed, long seedPosition, bool reverse)
{
    const int seedOffset = int(seed.offset);
    if (reverse)
    {
        const unsigned readLength = reads.at(seed.readIndex).length;
        return seedPosition + seed.length + seedOffset - readLength;
    }
    return seedPosition - seedOffset;
}

// FragmentBuilder.cpp:219-249
void FragmentBuilder::addMatch(const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds, const Match &match, const Cluster &cluster)
{
    const SeedId seedId(match.seedId);
    const unsigned seedIndex = unsigned(seedId.getSeed());
    const SeedMetadata &seedMetadata = seeds.at(seedIndex);
    const unsigned readIndex = seedMetadata.readIndex;
    const ReferencePosition seedLocation = ReferencePosition::fromValue(match.location);
    const bool reverse = seedId.isReverse();
    const long readPosition = getReadPosition(reads, seedMetadata, long(seedLocation.getPosition()), reverse);
    fragments[readIndex].push_back(FragmentMetadata(&cluster, 0, readIndex));
    FragmentMetadata &fragment = fragments[readIndex].back();
    fragment.firstSeedIndex = int(seedIndex);
    fragment.contigId = unsigned(seedLocation.getContigId());
    fragment.position = readPosition;
    fragment.reverse = reverse;
    if (seedMetadata.length != 64 /*STRONG_SEED_LENGTH*/ && seedLocation.hasNeighbors())
    {
        fragment.nonUniqueSeedOffsets.first = std::min<unsigned>(fragment.nonUniqueSeedOffsets.first, seedMetadata.offset);
        fragment.nonUniqueSeedOffsets.second = std::max<unsigned>(fragment.nonUniqueSeedOffsets.second, seedMetadata.offset);
    }
    else fragment.uniqueSeedCount = 1;
}

// FragmentBuilder.cpp:82-145
bool FragmentBuilder::build(const ContigList &contigs, const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds,
                            const Match *matchBegin, const Match *matchEnd, const Cluster &cluster, bool withGaps)
{
    clear();
    if (matchBegin < matchEnd)
    {
        for (; matchEnd != matchBegin && !ReferencePosition::fromValue(matchBegin->location).isNoMatch(); ++matchBegin)
        {
            const unsigned seed = unsigned(SeedId(matchBegin->seedId).getSeed());
            if (repeatThreshold > seedMatchCounts.at(seed))
            {
                if (ReferencePosition::fromValue(matchBegin->location).isTooManyMatch())
                {
                    seedMatchCounts[seed] = repeatThreshold;
                    ++repeatSeedsCount;
                }
                else if (repeatThreshold == ++seedMatchCounts[seed]) ++repeatSeedsCount;
                else addMatch(reads, seeds, *matchBegin, cluster);
            }
        }
        if (repeatSeedsCount)
        {
            for (unsigned r = 0; r < 2; ++r)
            {
                FragmentMetadataList &l = fragments[r];
                const std::vector<unsigned> &counts = seedMatchCounts; const unsigned thr = repeatThreshold;
                l.erase(std::remove_if(l.begin(), l.end(), [&](const FragmentMetadata &f) { return counts[f.firstSeedIndex] >= thr; }), l.end());
            }
        }
        if (!fragments[0].empty() || !fragments[1].empty())
        {
            alignFragments(contigs, reads, seeds, withGaps);
            return true;
        }
    }
    return false;
}

// FragmentBuilder.cpp:147-217
void FragmentBuilder::alignFragments(const ContigList &contigs, const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds, bool withGaps)
{
    for (unsigned r = 0; r < 2; ++r)
    {
        FragmentMetadataList &fragmentList = fragments[r];
        if (fragmentList.empty()) continue;
        consolidateDuplicateFragments(fragmentList, false);
        FragmentSequencingAdapterClipper adapterClipper(sequencingAdapters);      // one per read: FragmentBuilder.cpp:164
        for (size_t i = 0; i < fragmentList.size(); ++i)
        {
            FragmentMetadata &f = fragmentList[i];
            f.repeatSeedsCount = repeatSeedsCount;
            adapterClipper.checkInitStrand(f, contigs.at(f.contigId));
            ungappedAligner.alignUngapped(f, cigarBuffer, reads, adapterClipper, contigs.at(f.contigId));
        }
        consolidateDuplicateFragments(fragmentList, true);
        if (semialignedGapLimit)
        {
            simpleIndelAligner.alignSimpleIndels(cigarBuffer, contigs, reads, seeds, fragmentList);
            consolidateDuplicateFragments(fragmentList, true);
        }
        for (size_t i = 0; i < fragmentList.size(); ++i)
        {
            FragmentMetadata &f = fragmentList[i];
            adapterClipper.checkInitStrand(f, contigs[f.contigId]);       // (:195; every strand on the list was initialised by the loop above)
            if (withGaps && BandedSmithWaterman::mismatchesCutoff < f.mismatchCount)
            {
                FragmentMetadata tmp = f;
                const unsigned matchCount = gappedAligner.alignGapped(tmp, cigarBuffer, reads, adapterClipper, contigs[f.contigId]);
                if (matchCount && matchCount + BandedSmithWaterman::WIDEST_GAP_SIZE > f.getObservedLength() &&
                    (tmp.mismatchCount <= gappedMismatchesMax) && (f.mismatchCount > tmp.mismatchCount) &&
                    LP_LESS(f.logProbability, tmp.logProbability))
                    f = tmp;
            }
        }
        consolidateDuplicateFragments(fragmentList, true);
    }
}

// FragmentBuilder.cpp:279-324.  std::sort (libstdc++ introsort) is part of the behaviour: it is not stable for n > 16,
// and the survivor of a run of equal elements keeps its own firstSeedIndex/CIGAR.
void FragmentBuilder::consolidateDuplicateFragments(FragmentMetadataList &fragmentList, const bool removeUnaligned)
{
    std::sort(fragmentList.begin(), fragmentList.end());
    FragmentMetadataList::iterator lastFragment = fragmentList.begin();
    while (fragmentList.end() != lastFragment && removeUnaligned && !lastFragment->isAligned()) ++lastFragment;
    lastFragment = fragmentList.erase(fragmentList.begin(), lastFragment);
    if (2 > fragmentList.size()) return;
    for (FragmentMetadataList::iterator currentFragment = lastFragment + 1; fragmentList.end() != currentFragment; ++currentFragment)
    {
        if (removeUnaligned && !currentFragment->isAligned()) { }
        else if (*lastFragment == *currentFragment) lastFragment->consolidate(*currentFragment);
        else
        {
            ++lastFragment;
            if (lastFragment != currentFragment) *lastFragment = *currentFragment;
        }
    }
    fragmentList.resize(1 + lastFragment - fragmentList.begin());
}

} // namespace oracle
