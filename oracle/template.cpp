// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Template-length statistics, shadow rescue, template building / MAPQ, end clippers, per-tile match selection.
#include "oracle.hpp"
#include <cmath>
#include <algorithm>
#include <stdexcept>
#include <cassert>
#include <cstring>

namespace oracle
{

// ---------------------------------------------------------------- TemplateLengthStatistics
// include/alignment/TemplateLengthStatistics.hh:146-169
TemplateLengthStatistics::AlignmentModel TemplateLengthStatistics::alignmentModel(const FragmentMetadata &f1, const FragmentMetadata &f2)
{
    if (f1.getContigId() == f2.getContigId())
    {
        const unsigned positionMask = (f1.getPosition() <= f2.getPosition()) ? 0 : 4;
        return AlignmentModel(positionMask | (f1.isReverse() ? 2 : 0) | (f2.isReverse() ? 1 : 0));
    }
    return InvalidAlignmentModel;
}
unsigned long TemplateLengthStatistics::getLength(const FragmentMetadata &f1, const FragmentMetadata &f2)
{
    if (f1.getPosition() < f2.getPosition())
        return (unsigned long)std::max<long>(f2.getPosition() + f2.getObservedLength() - f1.getPosition(), f1.getObservedLength());
    return (unsigned long)std::max<long>(f1.getPosition() + f1.getObservedLength() - f2.getPosition(), f2.getObservedLength());
}
// TemplateLengthStatistics.hh:104-118
TemplateLengthStatistics::CheckModelResult TemplateLengthStatistics::checkModel(const FragmentMetadata &f1, const FragmentMetadata &f2) const
{
    if (f1.getContigId() == f2.getContigId())
    {
        const AlignmentModel model = alignmentModel(f1, f2);
        if (model == bestModels[0] || model == bestModels[1])
        {
            const unsigned long length = getLength(f1, f2);
            return (length > max) ? Oversized : (length < min) ? Undersized : Nominal;
        }
    }
    return NoMatch;
}
// TemplateLengthStatistics.cpp:67-77
bool TemplateLengthStatistics::matchModel(const FragmentMetadata &f1, const FragmentMetadata &f2) const
{
    const unsigned long length = getLength(f1, f2);
    const AlignmentModel model = alignmentModel(f1, f2);
    return (length <= max + TEMPLATE_LENGTH_THRESHOLD) && ((model == bestModels[0]) || (model == bestModels[1]));
}
// TemplateLengthStatistics.cpp:162-239
bool TemplateLengthStatistics::isValidModel(bool reverse, unsigned readIndex) const
{
    const unsigned shift = (readIndex + 1) % 2;
    return (reverse == bool((bestModels[0] >> shift) & 1)) || (reverse == bool((bestModels[1] >> shift) & 1));
}
bool TemplateLengthStatistics::firstFragment(bool reverse, unsigned readIndex) const
{
    const unsigned shift = (readIndex + 1) % 2;
    for (unsigned i = 0; 2 > i; ++i)
        if (reverse == bool((bestModels[i] >> shift) & 1)) return unsigned((bestModels[i] >> 2) & 1) == readIndex;
    return false;
}
bool TemplateLengthStatistics::mateOrientation(unsigned readIndex, bool reverse) const
{
    const unsigned shift = (readIndex + 1) % 2;
    for (unsigned i = 0; 2 > i; ++i)
        if (reverse == bool((bestModels[i] >> shift) & 1)) return (bestModels[i] >> readIndex) & 1;
    return (bestModels[0] >> readIndex) & 1;
}
long TemplateLengthStatistics::mateMinPosition(unsigned readIndex, bool reverse, long position, const unsigned *readLengths) const
{
    if (!isValidModel(reverse, readIndex)) return position;
    if (firstFragment(reverse, readIndex)) return position + mateMin - readLengths[(readIndex + 1) % 2];
    return position - mateMax + readLengths[readIndex];
}
long TemplateLengthStatistics::mateMaxPosition(unsigned readIndex, bool reverse, long position, const unsigned *readLengths) const
{
    if (!isValidModel(reverse, readIndex)) return position;
    if (firstFragment(reverse, readIndex)) return position + mateMax - readLengths[(readIndex + 1) % 2];
    return position - mateMin + readLengths[readIndex];
}

// TemplateLengthStatistics.cpp:31-38: boost::math::erf(3/sqrt2), erf(1/sqrt2)
static const double FRAGMENT_LENGTH_CONFIDENCE_INTERVAL = std::erf(3.0 / std::sqrt(2.0));
static const double FRAGMENT_LENGTH_CONFIDENCE_INTERVAL_1Z = std::erf(1.0 / std::sqrt(2.0));
static const double LOWER_PERCENT = (1.0 - FRAGMENT_LENGTH_CONFIDENCE_INTERVAL) / 2.0;
static const double UPPER_PERCENT = (1.0 + FRAGMENT_LENGTH_CONFIDENCE_INTERVAL) / 2.0;
static const double LOWER_PERCENT_1Z = (1.0 - FRAGMENT_LENGTH_CONFIDENCE_INTERVAL_1Z) / 2.0;
static const double UPPER_PERCENT_1Z = (1.0 + FRAGMENT_LENGTH_CONFIDENCE_INTERVAL_1Z) / 2.0;

void TemplateLengthDistribution::clear()
{
    stats.clear(); templateCount = 0; uniqueCount = 0; count = 0;
    for (size_t i = 0; i < histograms.size(); ++i) histograms[i].clear();
    lengthList.clear();
}

// TemplateLengthStatistics.cpp:105-159
void TemplateLengthDistribution::updateStatistics()
{
    typedef TemplateLengthStatistics TLS;
    const TLS oldStats = stats;
    stats.bestModels[0] = histograms[1].size() <= histograms[0].size() ? TLS::FFp : TLS::FRp;
    stats.bestModels[1] = TLS::AlignmentModel((stats.bestModels[0] + 1) % 2);
    for (size_t i = 2; histograms.size() > i; ++i)
    {
        if (histograms[i].size() > histograms[stats.bestModels[0]].size())
        {
            stats.bestModels[1] = stats.bestModels[0];
            stats.bestModels[0] = TLS::AlignmentModel(i);
        }
        else if (histograms[i].size() > histograms[stats.bestModels[1]].size()) stats.bestModels[1] = TLS::AlignmentModel(i);
    }
    lengthList.clear();
    lengthList.insert(lengthList.end(), histograms[stats.bestModels[0]].begin(), histograms[stats.bestModels[0]].end());
    lengthList.insert(lengthList.end(), histograms[stats.bestModels[1]].begin(), histograms[stats.bestModels[1]].end());
    std::sort(lengthList.begin(), lengthList.end());
    stats.setMin(lengthList.empty() ? 0 : lengthList[unsigned(lengthList.size() * LOWER_PERCENT)], mateDriftRange);
    stats.setMedian(lengthList.empty() ? TLS::TEMPLATE_LENGTH_THRESHOLD / 2 : lengthList[unsigned(lengthList.size() * 0.5)], mateDriftRange);
    stats.setMax(lengthList.empty() ? TLS::TEMPLATE_LENGTH_THRESHOLD : lengthList[unsigned(lengthList.size() * UPPER_PERCENT)], mateDriftRange);
    stats.lowStdDev = lengthList.empty() ? stats.median : (stats.median - lengthList[unsigned(lengthList.size() * LOWER_PERCENT_1Z)]);
    stats.highStdDev = lengthList.empty() ? stats.median : (lengthList[unsigned(lengthList.size() * UPPER_PERCENT_1Z)] - stats.median);
    if (oldStats.min == stats.min && oldStats.median == stats.median && oldStats.max == stats.max &&
        oldStats.lowStdDev == stats.lowStdDev && oldStats.highStdDev == stats.highStdDev &&
        oldStats.bestModels[0] == stats.bestModels[0] && oldStats.bestModels[1] == stats.bestModels[1])
        stats.stable = true;
}

// TemplateLengthStatistics.cpp:275-340
bool TemplateLengthDistribution::addTemplate(const std::vector<FragmentMetadataList> &fragments)
{
    typedef TemplateLengthStatistics TLS;
    if (fragments[0].empty() || fragments[1].empty()) return stats.stable;
    ++templateCount;
    if ((1 < fragments[0].size()) || (1 < fragments[1].size())) return stats.stable;
    ++uniqueCount;
    if (fragments[0][0].contigId != fragments[1][0].contigId) return stats.stable;
    const std::vector<uint32_t> &cigarBuffer = *fragments[0][0].cigarBuffer;
    for (size_t i = 0; 2 > i; ++i)
    {
        const unsigned firstOp = cigarBuffer[fragments[i][0].cigarOffset];
        const unsigned lastOp = cigarBuffer[fragments[i][0].cigarOffset + fragments[i][0].cigarLength - 1];
        if ((firstOp & 0xF) == INSERT || (lastOp & 0xF) == INSERT) return stats.stable;
    }
    const unsigned long length = TLS::getLength(fragments[0][0], fragments[1][0]);
    if (length > TLS::TEMPLATE_LENGTH_THRESHOLD) return stats.stable;
    const TLS::AlignmentModel am = TLS::alignmentModel(fragments[0][0], fragments[1][0]);
    if (TLS::InvalidAlignmentModel != am)
    {
        histograms[am].push_back(unsigned(length));
        ++count;
        if (0 == (count % UPDATE_FREQUENCY))
        {
            const TLS oldStats = stats;
            updateStatistics();
            if (oldStats.min == stats.min && oldStats.median == stats.median && oldStats.max == stats.max &&
                oldStats.lowStdDev == stats.lowStdDev && oldStats.highStdDev == stats.highStdDev)
                stats.stable = true;
        }
    }
    return stats.stable;
}

// TemplateLengthStatistics.cpp:342-358
bool TemplateLengthDistribution::finalize()
{
    const TemplateLengthStatistics oldStats = stats;
    updateStatistics();
    if (oldStats.min == stats.min && oldStats.median == stats.median && oldStats.max == stats.max &&
        oldStats.lowStdDev == stats.lowStdDev && oldStats.highStdDev == stats.highStdDev)
        stats.stable = true;
    return stats.stable;
}

// ---------------------------------------------------------------- RestOfGenomeCorrection (RestOfGenomeCorrection.hh:44-88)
RestOfGenomeCorrection::RestOfGenomeCorrection(const ContigList &contigs, const std::vector<ReadMetadata> &reads)
{
    const size_t gl = genomeLength(contigs);
    unsigned total = 0;
    rogCorrectionList[0] = rogCorrectionList[1] = 0;
    for (size_t i = 0; i < reads.size(); ++i)
    {
        rogCorrectionList[reads[i].index] = std::max(Quality::restOfGenomeCorrection(unsigned(gl), reads[i].length), std::numeric_limits<double>::min());
        total += reads[i].length;
    }
    rogCorrection = std::max(Quality::restOfGenomeCorrection(unsigned(gl), total), std::numeric_limits<double>::min());
}

// ---------------------------------------------------------------- BamTemplate (lib/alignment/BamTemplate.cpp)
void BamTemplate::initialize(const std::vector<ReadMetadata> &reads, const Cluster &cluster)
{
    fragments.clear(); alignmentScore = 0; properPair = false;
    for (size_t i = 0; i < reads.size(); ++i) fragments.push_back(FragmentMetadata(&cluster, cigarBuffer, reads[i].index));
}
bool BamTemplate::filterLowQualityFragments(unsigned mapqThreshold)
{
    bool ret = false; unsigned score = 0;
    for (unsigned i = 0; getFragmentCount() > i; ++i)
    {
        FragmentMetadata &fragment = getFragmentMetadata(i);
        if (mapqThreshold > fragment.getAlignmentScore())
        {
            fragment.cigarLength = 0; fragment.cigarOffset = 0; fragment.alignmentScore = 0;
            const FragmentMetadata &mate = getFragmentMetadata((i + 1) % getFragmentCount());
            fragment.position = mate.position; fragment.contigId = mate.contigId;
        }
        else if (fragment.isAligned()) ret = true;
        score += fragment.alignmentScore;
    }
    setAlignmentScore(score);
    return ret;
}

// ---------------------------------------------------------------- ShadowAligner (lib/alignment/ShadowAligner.cpp)
ShadowAligner::ShadowAligner(const Params &p)
    : gappedMismatchesMax(p.gappedMismatchesMax),
      ungappedAligner(p.gapMatchScore, p.gapMismatchScore, p.gapOpenScore, p.gapExtendScore, p.minGapExtendScore),
      gappedAligner(int(p.clusterLength()), p.gapMatchScore, p.gapMismatchScore, p.gapOpenScore, p.gapExtendScore, p.minGapExtendScore)
{
    for (const SequencingAdapterMetadata &m : p.adapters) sequencingAdapters.push_back(SequencingAdapter(m));
    shadowCigarBuffer.reserve(1 << 20);
}

// ShadowAligner.cpp:53-112
void ShadowAligner::findShadowCandidatePositions(const char *referenceBegin, const char *referenceEnd, const std::vector<char> &shadowSequence)
{
    shadowKmerPositions.assign(shadowKmerCount, -1);
    {
        KmerGenerator<unsigned> g(shadowSequence.data(), shadowSequence.data() + shadowSequence.size(), shadowKmerLength);
        unsigned kmer; const char *position;
        while (g.next(kmer, position)) if (-1 == shadowKmerPositions[kmer]) shadowKmerPositions[kmer] = short(position - shadowSequence.data());
    }
    KmerGenerator<unsigned> g(referenceBegin, referenceEnd, shadowKmerLength);
    unsigned kmer; const char *position;
    while (g.next(kmer, position))
    {
        if (-1 != shadowKmerPositions[kmer])
        {
            const long candidatePosition = position - referenceBegin - shadowKmerPositions[kmer];
            if (shadowCandidatePositions.empty() || shadowCandidatePositions.back() != candidatePosition)
            {
                if (shadowCandidatePositions.size() == candidatePositionsMax) break;
                shadowCandidatePositions.push_back(candidatePosition);
            }
        }
    }
    if (!shadowCandidatePositions.empty())
    {
        std::sort(shadowCandidatePositions.begin(), shadowCandidatePositions.end());
        shadowCandidatePositions.erase(std::unique(shadowCandidatePositions.begin(), shadowCandidatePositions.end()), shadowCandidatePositions.end());
    }
}

// ShadowAligner.cpp:119-149
static std::pair<long, long> calculateShadowRescueRange(const FragmentMetadata &orphan, const TemplateLengthStatistics &tls, const long bestTemplateLength)
{
    const Cluster &cluster = *orphan.cluster;
    const unsigned shadowReadIndex = (orphan.readIndex + 1) % 2;
    const unsigned readLengths[] = { cluster[0].getLength(), cluster[1].getLength() };
    long shadowMinPosition = tls.mateMinPosition(orphan.readIndex, orphan.reverse, orphan.position, readLengths);
    long shadowMaxPosition = tls.mateMaxPosition(orphan.readIndex, orphan.reverse, orphan.position, readLengths) + readLengths[shadowReadIndex] - 1;
    if (bestTemplateLength)
    {
        if (shadowMinPosition < long(orphan.getFStrandReferencePosition().getPosition()))
            shadowMinPosition = std::min(long(orphan.getRStrandReferencePosition().getPosition()) - bestTemplateLength, shadowMinPosition);
        if (shadowMaxPosition > long(orphan.getFStrandReferencePosition().getPosition()))
            shadowMaxPosition = std::max(long(orphan.getFStrandReferencePosition().getPosition()) + bestTemplateLength, shadowMaxPosition);
    }
    return std::make_pair(shadowMinPosition - 10, shadowMaxPosition + 10);
}

// ShadowAligner.cpp:155-291
bool ShadowAligner::rescueShadow(const ContigList &contigList, const FragmentMetadata &orphan, FragmentMetadataList &shadowList, size_t shadowListCapacity,
                                 const std::vector<ReadMetadata> &reads, const TemplateLengthStatistics &tls, const long bestTemplateLength)
{
    if (!tls.isCoherent()) return false;
    shadowCigarBuffer.clear();
    const Cluster &cluster = *orphan.cluster;
    const unsigned shadowReadIndex = (orphan.readIndex + 1) % 2;
    const Read &shadowRead = cluster[shadowReadIndex];
    const Contig &contig = contigList[orphan.contigId];
    const bool shadowReverse = tls.mateOrientation(orphan.readIndex, orphan.reverse);
    const std::vector<char> &reference = contig.forward;
    const std::pair<long, long> range = calculateShadowRescueRange(orphan, tls, bestTemplateLength);
    if (range.second < range.first) return false;
    if (range.second + 1 + long(shadowRead.getLength()) < 0) return false;
    shadowCandidatePositions.clear();
    const std::vector<char> &shadowSequence = shadowReverse ? shadowRead.reverseSequence : shadowRead.forwardSequence;
    const long candidatePositionOffset = std::max(0L, range.first);
    findShadowCandidatePositions(reference.data() + candidatePositionOffset,
                                 reference.data() + std::min((long)reference.size(), range.second + 1), shadowSequence);
    shadowList.clear();
    shadowList.reserve(shadowListCapacity); // pointers into the list must stay valid (the reference pre-reserves 1000)
    FragmentSequencingAdapterClipper adapterClipper(sequencingAdapters);       // ShadowAligner.cpp:207: one per rescue, the first candidate position decides
    FragmentMetadata *bestFragment = 0;
    for (size_t c = 0; c < shadowCandidatePositions.size(); ++c)
    {
        long strandPosition = shadowCandidatePositions[c];
        if (shadowList.size() == shadowListCapacity) return false;
        strandPosition += candidatePositionOffset;
        FragmentMetadata fragment(&cluster, &shadowCigarBuffer, shadowReadIndex);
        fragment.reverse = shadowReverse;
        fragment.contigId = orphan.contigId;
        fragment.position = strandPosition;
        adapterClipper.checkInitStrand(fragment, contig);
        if (ungappedAligner.alignUngapped(fragment, shadowCigarBuffer, reads, adapterClipper, contig))
        {
            shadowList.push_back(fragment);
            if (0 == bestFragment || LP_LESS(bestFragment->logProbability, fragment.logProbability)) bestFragment = &shadowList.back();
        }
    }
    if (!bestFragment) return false;
    if (BandedSmithWaterman::mismatchesCutoff < bestFragment->mismatchCount)
    {
        for (size_t i = 0; i < shadowList.size(); ++i)
        {
            FragmentMetadata &fragment = shadowList[i];
            if (i + 1 != shadowList.size() && shadowList[i + 1].position - fragment.position < long(BandedSmithWaterman::distanceCutoff))
            {
                if (BandedSmithWaterman::mismatchesCutoff < fragment.mismatchCount)
                {
                    FragmentMetadata tmp = fragment;
                    const unsigned matchCount = gappedAligner.alignGapped(tmp, shadowCigarBuffer, reads, adapterClipper, contig);
                    if (matchCount && matchCount + BandedSmithWaterman::WIDEST_GAP_SIZE > fragment.getObservedLength() &&
                        (tmp.mismatchCount <= gappedMismatchesMax) && (fragment.mismatchCount > tmp.mismatchCount) &&
                        LP_LESS(fragment.logProbability, tmp.logProbability))
                    {
                        fragment = tmp;
                        if (LP_LESS(bestFragment->logProbability, fragment.logProbability)) bestFragment = &fragment;
                    }
                }
            }
        }
    }
    if (&shadowList.front() != bestFragment) std::swap(shadowList.front(), *bestFragment);
    return true;
}

// ---------------------------------------------------------------- TemplateBuilder (lib/alignment/TemplateBuilder.cpp)
static const double LOG_MISMATCH_Q40 = Quality::getLogMismatch(40);
static const double orphanLogProbabilitySlack = 100.0;

// TemplateBuilder.cpp:52-58
static bool isVeryBadAlignment(const FragmentMetadata &fragment)
{
    return fragment.matchesInARow < 32 &&
        (fragment.mismatchCount > fragment.getMappedLength() / 8 || fragment.logProbability < LOG_MISMATCH_Q40 / 4 * fragment.getMappedLength());
}

void TemplateBuilder::BestPairInfo::clear()
{
    bestTemplateLogProbability = -std::numeric_limits<double>::max(); bestTemplateScore = (unsigned long)(-1);
    resolvedTemplateCount = 0; bestPairEditDistance = 0; totalTemplateProbability = 0.0;
    bestPairFragments[0].clear(); bestPairFragments[1].clear();
}
long TemplateBuilder::BestPairInfo::getBestTemplateLength() const
{
    if (!resolvedTemplateCount) return 0;
    const ReferencePosition templateStart = std::min(bestPairFragments[0][0]->getFStrandReferencePosition(), bestPairFragments[1][0]->getFStrandReferencePosition());
    const ReferencePosition templateEnd = std::max(bestPairFragments[0][0]->getRStrandReferencePosition(), bestPairFragments[1][0]->getRStrandReferencePosition());
    if (templateEnd.getContigId() != templateStart.getContigId()) throw std::logic_error("Contigs must match");
    return long(templateEnd.getPosition()) - long(templateStart.getPosition());
}

TemplateBuilder::TemplateBuilder(const Params &p)
    : scatterRepeats(p.scatterRepeats), dodgyAlignmentScore(p.dodgyAlignmentScore), fragmentBuilder(p), bamTemplate(fragmentBuilder.cigarBuffer), shadowAligner(p)
{
    cigarBuffer.reserve(1 << 16);
}

// TemplateBuilder.cpp:97-125
bool TemplateBuilder::buildTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                    const Cluster &cluster, const TemplateLengthStatistics &tls, const unsigned mapqThreshold)
{
    bool ret = buildTemplate(contigs, rog, reads, fragmentBuilder.fragments, cluster, tls);
    if (ret && bamTemplate.hasAlignmentScore())
    {
        if (!bamTemplate.isProperPair()) ret = bamTemplate.filterLowQualityFragments(mapqThreshold);
        else if (mapqThreshold > bamTemplate.getAlignmentScore()) { bamTemplate.filterLowQualityFragments(-1U); ret = false; }
    }
    return ret;
}
// TemplateBuilder.cpp:126-174
bool TemplateBuilder::buildTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                    const std::vector<FragmentMetadataList> &fragments, const Cluster &cluster, const TemplateLengthStatistics &tls)
{
    cigarBuffer.clear();
    bamTemplate.initialize(reads, cluster);
    if (2 == reads.size() && 2 == fragments.size())
    {
        if (!fragments[0].empty() && !fragments[1].empty()) return pickBestPair(contigs, rog, reads, fragments, tls);
        else if (!fragments[0].empty() || !fragments[1].empty()) return rescueShadow(contigs, rog, reads, fragments, tls);
        return false;
    }
    else if (1 == reads.size() || 1 == fragments.size())
    {
        if (!fragments[0].empty()) return pickBestFragment(rog, tls, fragments[0]);
        return false;
    }
    throw std::logic_error("TemplateBuilder supports at most 2 reads");
}

// TemplateBuilder.cpp:177-226
TemplateBuilder::FragmentIterator TemplateBuilder::getBestFragment(const FragmentMetadataList &fragmentList) const
{
    std::vector<FragmentIterator> bestFragments;
    unsigned bestFragmentScore = -1U;
    double bestFragmentLogProbability = -std::numeric_limits<double>::max();
    for (FragmentIterator it = fragmentList.data(); fragmentList.data() + fragmentList.size() != it; ++it)
    {
        if (bestFragmentScore > it->smithWatermanScore ||
            (bestFragmentScore == it->smithWatermanScore && LP_LESS(bestFragmentLogProbability, it->logProbability)))
        {
            bestFragmentScore = it->smithWatermanScore; bestFragmentLogProbability = it->logProbability;
            bestFragments.clear(); bestFragments.push_back(it);
        }
        else if (bestFragmentScore == it->smithWatermanScore && LP_EQUALS(bestFragmentLogProbability, it->logProbability)) bestFragments.push_back(it);
    }
    const unsigned clusterId = unsigned(fragmentList[0].cluster->id);
    const unsigned repeatIndex = scatterRepeats ? (clusterId % bestFragments.size()) : 0;
    return bestFragments[repeatIndex];
}

// TemplateBuilder.cpp:233-285
bool TemplateBuilder::updateMappingScore(FragmentMetadata &fragment, const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &,
                                         const FragmentIterator listFragment, const FragmentMetadataList &fragmentList, const bool forceWellAnchored) const
{
    if (forceWellAnchored || fragment.isWellAnchored())
    {
        double neighborProbability = rog.getReadRogCorrection(listFragment->getReadIndex());
        for (FragmentIterator i = fragmentList.data(); fragmentList.data() + fragmentList.size() != i; ++i)
            if (listFragment != i) neighborProbability += exp(i->logProbability);
        fragment.alignmentScore = unsigned(floor(-10.0 * log10(neighborProbability / (neighborProbability + exp(listFragment->logProbability)))));
        return true;
    }
    fragment.alignmentScore = 0;
    return false;
}

// TemplateBuilder.cpp:287-391
void TemplateBuilder::locateBestPair(const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls, BestPairInfo &ret) const
{
    const FragmentIterator begin[2] = { fragments[0].data(), fragments[1].data() };
    const FragmentIterator end[2] = { begin[0] + fragments[0].size(), begin[1] + fragments[1].size() };
    ret.init(begin[0], begin[1]);
    FragmentIterator contigBegin[2] = { begin[0], begin[1] };
    FragmentIterator contigEnd[2];
    while ((end[0] != contigBegin[0]) && (end[1] != contigBegin[1]))
    {
        for (size_t i = 0; 2 > i; ++i)
        {
            contigEnd[i] = contigBegin[i] + 1;
            while ((end[i] != contigEnd[i]) && (contigEnd[i]->contigId == contigBegin[i]->contigId)) ++contigEnd[i];
        }
        if (contigBegin[0]->contigId == contigBegin[1]->contigId)
        {
            FragmentIterator currentFragment[2] = { contigBegin[0], contigBegin[1] };
            while (contigEnd[0] != currentFragment[0])
            {
                currentFragment[1] = contigBegin[1];
                while (contigEnd[1] != currentFragment[1])
                {
                    if (tls.matchModel(*currentFragment[0], *currentFragment[1]))
                    {
                        const double currentLogProbability = currentFragment[0]->logProbability + currentFragment[1]->logProbability;
                        const double currentProbability = exp(currentLogProbability);
                        const unsigned long templateScore = currentFragment[0]->smithWatermanScore + currentFragment[1]->smithWatermanScore;
                        ret.totalTemplateProbability += currentProbability;
                        if (0 == ret.resolvedTemplateCount || ret.bestTemplateScore > templateScore ||
                            (templateScore == ret.bestTemplateScore && LP_LESS(ret.bestTemplateLogProbability, currentLogProbability)))
                        {
                            ret.bestPairFragments[0].clear(); ret.bestPairFragments[1].clear();
                            ret.bestPairFragments[0].push_back(currentFragment[0]); ret.bestPairFragments[1].push_back(currentFragment[1]);
                            ret.bestTemplateScore = templateScore; ret.bestTemplateLogProbability = currentLogProbability;
                        }
                        else if (templateScore == ret.bestTemplateScore && LP_EQUALS(currentLogProbability, ret.bestTemplateLogProbability))
                        {
                            ret.bestPairFragments[0].push_back(currentFragment[0]); ret.bestPairFragments[1].push_back(currentFragment[1]);
                        }
                        ++ret.resolvedTemplateCount;
                    }
                    ++currentFragment[1];
                }
                ++currentFragment[0];
            }
            for (size_t i = 0; 2 > i; ++i) contigBegin[i] = contigEnd[i];
        }
        else
        {
            const size_t i = contigBegin[0]->contigId < contigBegin[1]->contigId ? 0 : 1;
            contigBegin[i] = contigEnd[i];
        }
    }
    if (ret.resolvedTemplateCount)
        ret.bestPairEditDistance = ret.bestPairFragments[0][0]->getEditDistance() + ret.bestPairFragments[1][0]->getEditDistance();
}

// TemplateBuilder.cpp:398-465
bool TemplateBuilder::buildPairedEndTemplate(const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls,
                                             const std::vector<FragmentMetadataList> &fragments, BestPairInfo &best)
{
    FragmentMetadata &read1 = bamTemplate.getFragmentMetadata(0);
    FragmentMetadata &read2 = bamTemplate.getFragmentMetadata(1);
    if (scatterRepeats)
    {
        const unsigned repeatIndex = unsigned(read1.cluster->id % best.bestPairFragments[0].size());
        std::swap(best.bestPairFragments[0][0], best.bestPairFragments[0][repeatIndex]);
        std::swap(best.bestPairFragments[1][0], best.bestPairFragments[1][repeatIndex]);
    }
    read1 = *best.bestPairFragments[0][0];
    read2 = *best.bestPairFragments[1][0];
    const bool r1WellAnchored = updateMappingScore(read1, rog, tls, best.bestPairFragments[0][0], fragments[0], read2.isWellAnchored());
    const bool r2WellAnchored = updateMappingScore(read2, rog, tls, best.bestPairFragments[1][0], fragments[1], read1.isWellAnchored());
    bamTemplate.setProperPair(TemplateLengthStatistics::Nominal == tls.checkModel(read1, read2));
    if (r1WellAnchored || r2WellAnchored)
    {
        const double otherPairsProbability = (best.totalTemplateProbability - exp(best.bestTemplateLogProbability)) + rog.getRogCorrection();
        bamTemplate.setAlignmentScore(unsigned(floor(-10.0 * log10(otherPairsProbability / (best.totalTemplateProbability + rog.getRogCorrection())))));
        return r1WellAnchored && r2WellAnchored && !read1.repeatSeedsCount && !read2.repeatSeedsCount;
    }
    bamTemplate.setAlignmentScore(-1U);
    return false;
}

// TemplateBuilder.cpp:467-493, :1010-1033
bool TemplateBuilder::flagDodgyTemplate(FragmentMetadata &orphan, FragmentMetadata &shadow, BamTemplate &t) const
{
    if (DODGY_ALIGNMENT_SCORE_UNALIGNED == dodgyAlignmentScore) { orphan.setNoMatch(); shadow.setNoMatch(); t.setAlignmentScore(-1U); return false; }
    orphan.alignmentScore = -1U; shadow.alignmentScore = -1U; t.setAlignmentScore(-1U);
    return true;
}
bool TemplateBuilder::flagDodgyTemplate(FragmentMetadata &orphan, BamTemplate &t) const
{
    if (DODGY_ALIGNMENT_SCORE_UNALIGNED == dodgyAlignmentScore) { orphan.setNoMatch(); t.setAlignmentScore(-1U); return false; }
    orphan.alignmentScore = -1U; t.setAlignmentScore(-1U);
    return true;
}

// TemplateBuilder.cpp:678-714
FragmentMetadata TemplateBuilder::cloneWithCigar(const FragmentMetadata &right)
{
    FragmentMetadata ret = right;
    ret.cigarBuffer = &cigarBuffer;
    ret.cigarOffset = unsigned(cigarBuffer.size());
    std::vector<uint32_t> tmp(right.cigarBuffer->begin() + right.cigarOffset, right.cigarBuffer->begin() + right.cigarOffset + right.cigarLength);
    cigarBuffer.insert(cigarBuffer.end(), tmp.begin(), tmp.end());
    return ret;
}
double TemplateBuilder::sumUniqueShadowProbabilities(std::vector<ShadowProbability> &v)
{
    double ret = 0.0;
    std::sort(v.begin(), v.end());
    // std::unique_copy keeps the first element of each run of elements equal to their predecessor-kept element
    for (size_t i = 0; i < v.size();)
    {
        ret += exp(v[i].logProbability);
        size_t j = i + 1;
        while (j < v.size() && v[i] == v[j]) ++j;
        i = j;
    }
    return ret;
}
double TemplateBuilder::sumUniquePairProbabilities(std::vector<PairProbability> &v)
{
    double ret = 0.0;
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size();)
    {
        ret += exp(v[i].logProbability());
        size_t j = i + 1;
        while (j < v.size() && v[i] == v[j]) ++j;
        i = j;
    }
    return ret;
}

// TemplateBuilder.cpp:495-676
bool TemplateBuilder::rescueShadow(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                   const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls)
{
    const unsigned orphanIndex = fragments[0].empty() ? 1 : 0;
    const unsigned shadowIndex = (orphanIndex + 1) % 2;
    const FragmentIterator bestOrphanIterator = getBestFragment(fragments[orphanIndex]);
    BestPairInfo &bestPair = bestRescuedPair;
    bestPair.clear();
    bestPair.bestPairFragments[orphanIndex].push_back(bestOrphanIterator);
    allShadowProbabilities[orphanIndex].clear();
    for (FragmentIterator orphanIterator = fragments[orphanIndex].data(); fragments[orphanIndex].data() + fragments[orphanIndex].size() != orphanIterator; ++orphanIterator)
    {
        const FragmentMetadata &orphan = *orphanIterator;
        shadowList.clear();
        if (LP_LESS(orphan.logProbability + orphanLogProbabilitySlack, bestOrphanIterator->logProbability)) { }
        else
        {
            ++rescueCalls;
            if (shadowAligner.rescueShadow(contigs, orphan, shadowList, TRACKED_REPEATS_MAX_ONE_READ, reads, tls, 0))
            {
                const FragmentMetadata &bestRescued = shadowList.front();
                const double currentTemplateLogProbability = orphan.logProbability + bestRescued.logProbability;
                const unsigned long templateScore = orphan.smithWatermanScore + bestRescued.smithWatermanScore;
                if (!isVeryBadAlignment(bestRescued))
                {
                    if (0 == bestPair.resolvedTemplateCount || templateScore < bestPair.bestTemplateScore ||
                        (templateScore == bestPair.bestTemplateScore && LP_LESS(bestPair.bestTemplateLogProbability, currentTemplateLogProbability)))
                    {
                        bestPair.bestTemplateLogProbability = currentTemplateLogProbability; bestPair.bestTemplateScore = templateScore;
                        bestPair.bestPairFragments[orphanIndex].clear(); bestPair.bestPairFragments[orphanIndex].push_back(orphanIterator);
                        bestOrphanShadows[orphanIndex].clear(); bestOrphanShadows[orphanIndex].push_back(cloneWithCigar(bestRescued));
                    }
                    else if (templateScore == bestPair.bestTemplateScore && LP_EQUALS(currentTemplateLogProbability, bestPair.bestTemplateLogProbability))
                    {
                        bestPair.bestPairFragments[orphanIndex].push_back(orphanIterator);
                        bestOrphanShadows[orphanIndex].push_back(cloneWithCigar(bestRescued));
                    }
                    ++bestPair.resolvedTemplateCount;
                }
            }
            rescueCandidates += shadowAligner.shadowCandidatePositions.size();
        }
        for (size_t s = 0; s < shadowList.size(); ++s)
        {
            allShadowProbabilities[orphanIndex].push_back(ShadowProbability(shadowList[s]));
            bestPair.totalTemplateProbability += exp(orphan.logProbability + shadowList[s].logProbability);
        }
    }
    const double totalShadowProbability = (0 < bestPair.resolvedTemplateCount) ? sumUniqueShadowProbabilities(allShadowProbabilities[orphanIndex]) : 0.0;
    bool ret = true;
    FragmentMetadata &orphan = bamTemplate.getFragmentMetadata(orphanIndex);
    if (0 < bestPair.resolvedTemplateCount)
    {
        const unsigned clusterId = unsigned(fragments[orphanIndex][0].cluster->id);
        const unsigned repeatIndex = scatterRepeats ? clusterId % unsigned(bestPair.bestPairFragments[orphanIndex].size()) : 0;
        orphan = *bestPair.bestPairFragments[orphanIndex][repeatIndex];
        FragmentMetadata &bestShadow = bestOrphanShadows[orphanIndex][repeatIndex];
        const bool assumeWellAnchored = updateMappingScore(orphan, rog, tls, bestPair.bestPairFragments[orphanIndex][repeatIndex], fragments[orphanIndex],
                                                           0 == orphan.getEditDistance() + bestShadow.getEditDistance());
        if (assumeWellAnchored)
        {
            const double shadowRog = rog.getReadRogCorrection(bestShadow.getReadIndex());
            const double otherShadowsProbability = (totalShadowProbability - exp(bestShadow.logProbability)) + shadowRog;
            bestShadow.alignmentScore = unsigned(floor(-10.0 * log10(otherShadowsProbability / (totalShadowProbability + shadowRog))));
            const double otherPairsProbability = (bestPair.totalTemplateProbability - exp(bestPair.bestTemplateLogProbability)) + rog.getRogCorrection();
            bamTemplate.setAlignmentScore(unsigned(floor(-10.0 * log10(otherPairsProbability / (bestPair.totalTemplateProbability + rog.getRogCorrection())))));
            if (!orphan.alignmentScore || !orphan.isWellAnchored())
            {
                bamTemplate.setAlignmentScore(std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bamTemplate.getAlignmentScore()));
                bestShadow.alignmentScore = std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bestShadow.alignmentScore);
                orphan.alignmentScore = std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.alignmentScore);
            }
        }
        else ret = flagDodgyTemplate(orphan, bestShadow, bamTemplate);
        bamTemplate.getFragmentMetadata(shadowIndex) = bestShadow;
        bamTemplate.setProperPair(TemplateLengthStatistics::Nominal == tls.checkModel(orphan, bestShadow));
    }
    else
    {
        orphan = *bestOrphanIterator;
        FragmentMetadata &shadow = bamTemplate.getFragmentMetadata(shadowIndex);
        if (isVeryBadAlignment(orphan)) { orphan.setNoMatch(); shadow.setNoMatch(); ret = false; }
        else
        {
            shadow.contigId = orphan.contigId; shadow.position = orphan.position; shadow.readIndex = shadowIndex;
            shadow.alignmentScore = 0; shadow.cigarLength = 0;
            if (!updateMappingScore(orphan, rog, tls, bestOrphanIterator, fragments[orphanIndex], 0 == orphan.getEditDistance()))
                ret = flagDodgyTemplate(orphan, shadow, bamTemplate);
            else
            {
                if (!orphan.isWellAnchored()) orphan.setAlignmentScore(std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.getAlignmentScore()));
                bamTemplate.setAlignmentScore(0);
            }
        }
    }
    return ret;
}

// TemplateBuilder.cpp:716-866
bool TemplateBuilder::buildDisjoinedTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                             const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls, const BestPairInfo &knownBestPair)
{
    const FragmentIterator bestDisjoinedFragments[2] = { getBestFragment(fragments[0]), getBestFragment(fragments[1]) };
    const long bestTemplateLength = knownBestPair.getBestTemplateLength();
    unsigned bestOrphanIndex = 0;
    BestPairInfo &bestOrphans = bestRescuedPair;
    bestOrphans.init(bestDisjoinedFragments[0], bestDisjoinedFragments[1]);
    allPairProbabilities.clear();
    for (unsigned orphanIndex = 0; 2 > orphanIndex; ++orphanIndex)
    {
        allShadowProbabilities[orphanIndex].clear();
        bestOrphanShadows[orphanIndex].clear();
        for (FragmentIterator orphanIterator = fragments[orphanIndex].data(); fragments[orphanIndex].data() + fragments[orphanIndex].size() != orphanIterator; ++orphanIterator)
        {
            const FragmentMetadata &orphan = *orphanIterator;
            const bool skipThisOrphan = (knownBestPair.resolvedTemplateCount ?
                orphan.getEditDistance() > (knownBestPair.bestPairEditDistance + SKIP_ORPHAN_EDIT_DISTANCE) :
                LP_LESS(orphan.logProbability + orphanLogProbabilitySlack, bestDisjoinedFragments[orphanIndex]->logProbability));
            shadowList.clear();
            bool rescued = false;
            if (!skipThisOrphan)
            {
                ++rescueCalls;
                rescued = shadowAligner.rescueShadow(contigs, orphan, shadowList, TRACKED_REPEATS_MAX_ONE_READ, reads, tls, bestTemplateLength);
                rescueCandidates += shadowAligner.shadowCandidatePositions.size();
            }
            if (rescued)
            {
                const FragmentMetadata &bestRescued = shadowList.front();
                const double currentTemplateLogProbability = orphan.logProbability + bestRescued.logProbability;
                const unsigned rescuedEditDistance = orphan.getEditDistance() + bestRescued.getEditDistance();
                if (isVeryBadAlignment(bestRescued)) { }
                else if (!knownBestPair.resolvedTemplateCount || (knownBestPair.bestPairEditDistance + SKIP_ORPHAN_EDIT_DISTANCE) >= rescuedEditDistance)
                {
                    const unsigned long templateScore = orphan.smithWatermanScore + bestRescued.smithWatermanScore;
                    if (0 == bestOrphans.resolvedTemplateCount || templateScore < bestOrphans.bestTemplateScore ||
                        (templateScore == bestOrphans.bestTemplateScore && LP_LESS(bestOrphans.bestTemplateLogProbability, currentTemplateLogProbability)))
                    {
                        bestOrphans.bestTemplateLogProbability = currentTemplateLogProbability; bestOrphans.bestTemplateScore = templateScore;
                        bestOrphans.bestPairFragments[orphanIndex].clear(); bestOrphans.bestPairFragments[orphanIndex].push_back(orphanIterator);
                        bestOrphanShadows[orphanIndex].clear(); bestOrphanShadows[orphanIndex].push_back(cloneWithCigar(bestRescued));
                        bestOrphanIndex = orphanIndex;
                    }
                    else if (templateScore == bestOrphans.bestTemplateScore && LP_EQUALS(currentTemplateLogProbability, bestOrphans.bestTemplateLogProbability))
                    {
                        bestOrphans.bestPairFragments[orphanIndex].push_back(orphanIterator);
                        bestOrphanShadows[orphanIndex].push_back(cloneWithCigar(bestRescued));
                    }
                    ++bestOrphans.resolvedTemplateCount;
                }
            }
            for (size_t s = 0; s < shadowList.size(); ++s)
            {
                const FragmentMetadata &shadow = shadowList[s];
                allPairProbabilities.push_back(PairProbability(0 == orphanIndex ? orphan : shadow, 0 == orphanIndex ? shadow : orphan));
                allShadowProbabilities[orphanIndex].push_back(ShadowProbability(shadow));
            }
        }
    }
    const unsigned bestShadowIndex = (bestOrphanIndex + 1) % 2;
    double totalShadowProbability = 0.0, totalOrphanProbability = 0.0;
    if (0 < bestOrphans.resolvedTemplateCount)
    {
        for (size_t i = 0; i < fragments[bestShadowIndex].size(); ++i) allShadowProbabilities[bestOrphanIndex].push_back(ShadowProbability(fragments[bestShadowIndex][i]));
        totalShadowProbability = sumUniqueShadowProbabilities(allShadowProbabilities[bestOrphanIndex]);
        for (size_t i = 0; i < fragments[bestOrphanIndex].size(); ++i) allShadowProbabilities[bestShadowIndex].push_back(ShadowProbability(fragments[bestOrphanIndex][i]));
        totalOrphanProbability = sumUniqueShadowProbabilities(allShadowProbabilities[bestShadowIndex]);
        bestOrphans.totalTemplateProbability += sumUniquePairProbabilities(allPairProbabilities);
    }
    return scoreDisjoinedTemplate(fragments, rog, tls, bestOrphans, knownBestPair, bestOrphanIndex, totalShadowProbability, totalOrphanProbability, bestDisjoinedFragments);
}

// TemplateBuilder.cpp:868-1008
bool TemplateBuilder::scoreDisjoinedTemplate(const std::vector<FragmentMetadataList> &fragments, const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls,
                                             const BestPairInfo &bestOrphans, const BestPairInfo &knownBestPair, const unsigned bestOrphanIndex,
                                             const double totalShadowProbability, const double totalOrphanProbability, const FragmentIterator bestDisjoinedFragments[2])
{
    bool ret = true;
    if (0 < bestOrphans.resolvedTemplateCount)
    {
        const unsigned clusterId = unsigned(fragments[0][0].cluster->id);
        const unsigned repeatIndex = scatterRepeats ? clusterId % unsigned(bestOrphans.bestPairFragments[bestOrphanIndex].size()) : 0;
        const FragmentMetadata &bestOrphan = *bestOrphans.bestPairFragments[bestOrphanIndex][repeatIndex];
        FragmentMetadata &bestShadow = bestOrphanShadows[bestOrphanIndex][repeatIndex];
        const bool rediscovered = !repeatIndex && knownBestPair.resolvedTemplateCount &&
            *knownBestPair.bestPairFragments[bestOrphan.getReadIndex()][0] == bestOrphan &&
            *knownBestPair.bestPairFragments[bestShadow.getReadIndex()][0] == bestShadow;
        FragmentMetadata &orphan = bamTemplate.getFragmentMetadata(bestOrphan.getReadIndex());
        orphan = bestOrphan;
        const bool shadowWellAnchored = rediscovered && knownBestPair.bestPairFragments[bestShadow.getReadIndex()][0]->isWellAnchored();
        const bool assumeWellAnchored = updateMappingScore(orphan, rog, tls, bestOrphans.bestPairFragments[bestOrphan.getReadIndex()][repeatIndex],
                                                           fragments[bestOrphan.getReadIndex()],
                                                           0 == orphan.getEditDistance() + bestShadow.getEditDistance() || shadowWellAnchored);
        bamTemplate.setProperPair(TemplateLengthStatistics::Nominal == tls.checkModel(orphan, bestShadow));
        if (assumeWellAnchored)
        {
            const double shadowRog = rog.getReadRogCorrection(bestShadow.getReadIndex());
            const double otherShadowsProbability = (totalShadowProbability - exp(bestShadow.logProbability)) + shadowRog;
            bestShadow.alignmentScore = unsigned(floor(-10.0 * log10(otherShadowsProbability / (totalShadowProbability + shadowRog))));
            const double orphanRog = rog.getReadRogCorrection(bestOrphan.getReadIndex());
            const double otherOrphansProbability = (totalOrphanProbability - exp(bestOrphan.logProbability)) + orphanRog;
            orphan.alignmentScore = unsigned(floor(-10.0 * log10(otherOrphansProbability / (totalOrphanProbability + orphanRog))));
            const double otherPairsProbability = (bestOrphans.totalTemplateProbability - exp(bestOrphans.bestTemplateLogProbability)) + rog.getRogCorrection();
            bamTemplate.setAlignmentScore(unsigned(floor(-10.0 * log10(otherPairsProbability / (bestOrphans.totalTemplateProbability + rog.getRogCorrection())))));
            if ((!orphan.alignmentScore || !orphan.isWellAnchored()) && (!bestShadow.alignmentScore || !shadowWellAnchored))
            {
                bamTemplate.setAlignmentScore(std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bamTemplate.getAlignmentScore()));
                bestShadow.alignmentScore = std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, bestShadow.alignmentScore);
                orphan.alignmentScore = std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, orphan.alignmentScore);
            }
            bamTemplate.getFragmentMetadata(bestShadow.getReadIndex()) = bestShadow;
        }
        else
        {
            ret = flagDodgyTemplate(orphan, bestShadow, bamTemplate);
            bamTemplate.getFragmentMetadata(bestShadow.getReadIndex()) = bestShadow;
        }
    }
    else if (knownBestPair.resolvedTemplateCount)
        ret = flagDodgyTemplate(bamTemplate.getFragmentMetadata(0), bamTemplate.getFragmentMetadata(1), bamTemplate);
    else
    {
        FragmentMetadata &read1 = bamTemplate.getFragmentMetadata(0);
        FragmentMetadata &read2 = bamTemplate.getFragmentMetadata(1);
        read1 = *bestDisjoinedFragments[0];
        read2 = *bestDisjoinedFragments[1];
        bamTemplate.setAlignmentScore(0);
        bamTemplate.setProperPair(false);
        const bool a1 = updateMappingScore(read1, rog, tls, bestDisjoinedFragments[0], fragments[0], 0 == read1.getEditDistance());
        const bool a2 = updateMappingScore(read2, rog, tls, bestDisjoinedFragments[1], fragments[1], 0 == read2.getEditDistance());
        if (!a1 && !a2) ret = flagDodgyTemplate(read1, read2, bamTemplate);
        else
        {
            if (!read1.isWellAnchored()) read1.setAlignmentScore(std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, read1.getAlignmentScore()));
            if (!read2.isWellAnchored()) read2.setAlignmentScore(std::min(DODGY_BUT_CLEAN_ALIGNMENT_SCORE, read2.getAlignmentScore()));
        }
    }
    return ret;
}

// TemplateBuilder.cpp:1035-1058
bool TemplateBuilder::pickBestFragment(const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls, const FragmentMetadataList &fragmentList)
{
    if (!fragmentList.empty())
    {
        const FragmentIterator bestFragment = getBestFragment(fragmentList);
        bamTemplate.getFragmentMetadata(0) = *bestFragment;
        if (!updateMappingScore(bamTemplate.getFragmentMetadata(0), rog, tls, bestFragment, fragmentList, false))
            return flagDodgyTemplate(bamTemplate.getFragmentMetadata(0), bamTemplate);
        return true;
    }
    return false;
}

// TemplateBuilder.cpp:1060-1086
bool TemplateBuilder::pickBestPair(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                   const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls)
{
    locateBestPair(fragments, tls, bestCombinationPairInfo);
    if (!bestCombinationPairInfo.resolvedTemplateCount ||
        !buildPairedEndTemplate(rog, tls, fragments, bestCombinationPairInfo) ||
        bestCombinationPairInfo.bestPairEditDistance)
        return buildDisjoinedTemplate(contigs, rog, reads, fragments, tls, bestCombinationPairInfo);
    return true;
}

// ---------------------------------------------------------------- clippers
// include/alignment/Alignment.hh:55-88
template <typename SeqIt, typename RefIt>
static std::pair<unsigned, unsigned> clipMismatches(SeqIt sequenceBegin, const SeqIt sequenceEnd, RefIt referenceBegin, RefIt referenceEnd, const unsigned CONSECUTIVE_MATCHES_MIN)
{
    unsigned matchesInARow = 0, ediDistanceMismatches = 0, ediDistanceMismatchesUnclipped = 0, ret = 0;
    while (sequenceEnd != sequenceBegin && referenceBegin != referenceEnd && CONSECUTIVE_MATCHES_MIN > matchesInARow)
    {
        const char sequenceBase = *sequenceBegin;
        if (isMatch(sequenceBase, *referenceBegin)) { ++matchesInARow; ediDistanceMismatchesUnclipped += (sequenceBase != *referenceBegin); }
        else { matchesInARow = 0; ediDistanceMismatchesUnclipped = 0; }
        ediDistanceMismatches += (sequenceBase != *referenceBegin);
        ++sequenceBegin; ++referenceBegin; ++ret;
    }
    return (CONSECUTIVE_MATCHES_MIN == matchesInARow) ? std::make_pair(ret - matchesInARow, ediDistanceMismatches - ediDistanceMismatchesUnclipped) : std::make_pair(0U, 0U);
}

// matchSelector/SemialignedEndsClipper.cpp:31-91
bool SemialignedEndsClipper::clipLeftSide(const ContigList &contigList, FragmentMetadata &f)
{
    const Read &read = f.getRead();
    const char *sequenceBegin = read.getStrandSequence(f.reverse).data();
    unsigned oldCigarOffset = f.cigarOffset, oldCigarLength = f.cigarLength;
    std::pair<unsigned, CigarOp> operation = cigarDecode(f.cigarBuffer->at(oldCigarOffset));
    unsigned softClippedBeginBases = 0;
    if (SOFT_CLIP == operation.second)
    {
        if (2 > f.cigarLength) return false;
        ++oldCigarOffset; --oldCigarLength;
        softClippedBeginBases = operation.first;
        sequenceBegin += operation.first;
        operation = cigarDecode(f.cigarBuffer->at(oldCigarOffset));
    }
    if (ALIGN == operation.second)
    {
        unsigned mappedBeginBases = operation.first;
        const char *sequenceEnd = sequenceBegin + mappedBeginBases;
        const std::vector<char> &reference = contigList.at(f.contigId).forward;
        const char *referenceBegin = reference.data() + f.position;
        const std::pair<unsigned, unsigned> clipped = clipMismatches(sequenceBegin, sequenceEnd, referenceBegin, reference.data() + reference.size(), CONSECUTIVE_MATCHES_MIN);
        if (clipped.first)
        {
            const std::vector<uint32_t> old(f.cigarBuffer->begin() + oldCigarOffset + 1, f.cigarBuffer->begin() + oldCigarOffset + oldCigarLength);
            f.cigarOffset = unsigned(cigarBuffer.size());
            f.observedLength -= clipped.first;
            softClippedBeginBases += clipped.first;
            mappedBeginBases -= clipped.first;
            f.position += clipped.first;
            f.editDistance -= clipped.second;
            cigarBuffer.push_back(cigarEncode(softClippedBeginBases, SOFT_CLIP));
            cigarBuffer.push_back(cigarEncode(mappedBeginBases, ALIGN));
            cigarBuffer.insert(cigarBuffer.end(), old.begin(), old.end());
            f.cigarBuffer = &cigarBuffer;
            f.cigarLength = unsigned(cigarBuffer.size()) - f.cigarOffset;
            return true;
        }
    }
    return false;
}

// SemialignedEndsClipper.cpp:93-156
bool SemialignedEndsClipper::clipRightSide(const ContigList &contigList, FragmentMetadata &f)
{
    const Read &read = f.getRead();
    const std::vector<char> &seq = read.getStrandSequence(f.reverse);
    std::reverse_iterator<const char *> sequenceRBegin(seq.data() + seq.size());
    unsigned oldCigarOffset = f.cigarOffset, oldCigarLength = f.cigarLength;
    std::pair<unsigned, CigarOp> operation = cigarDecode(f.cigarBuffer->at(oldCigarOffset + oldCigarLength - 1));
    unsigned softClippedEndBases = 0;
    if (SOFT_CLIP == operation.second)
    {
        if (2 > f.cigarLength) return false;
        --oldCigarLength;
        softClippedEndBases = operation.first;
        sequenceRBegin += operation.first;
        operation = cigarDecode(f.cigarBuffer->at(oldCigarOffset + oldCigarLength - 1));
    }
    if (ALIGN == operation.second)
    {
        unsigned mappedEndBases = operation.first;
        std::reverse_iterator<const char *> sequenceREnd = sequenceRBegin + mappedEndBases;
        const std::vector<char> &reference = contigList.at(f.contigId).forward;
        std::reverse_iterator<const char *> referenceRBegin(reference.data() + f.position + f.getObservedLength());
        std::reverse_iterator<const char *> referenceREnd(reference.data());
        const std::pair<unsigned, unsigned> clipped = clipMismatches(sequenceRBegin, sequenceREnd, referenceRBegin, referenceREnd, CONSECUTIVE_MATCHES_MIN);
        if (clipped.first)
        {
            const std::vector<uint32_t> old(f.cigarBuffer->begin() + oldCigarOffset, f.cigarBuffer->begin() + oldCigarOffset + oldCigarLength - 1);
            f.cigarOffset = unsigned(cigarBuffer.size());
            f.observedLength -= clipped.first;
            softClippedEndBases += clipped.first;
            f.editDistance -= clipped.second;
            mappedEndBases -= clipped.first;
            cigarBuffer.insert(cigarBuffer.end(), old.begin(), old.end());
            cigarBuffer.push_back(cigarEncode(mappedEndBases, ALIGN));
            cigarBuffer.push_back(cigarEncode(softClippedEndBases, SOFT_CLIP));
            f.cigarBuffer = &cigarBuffer;
            f.cigarLength = unsigned(cigarBuffer.size()) - f.cigarOffset;
            return true;
        }
    }
    return false;
}
// SemialignedEndsClipper.cpp:161-205
bool SemialignedEndsClipper::clip(const ContigList &contigList, FragmentMetadata &f)
{
    if (!f.isAligned()) return false;
    bool ret = clipLeftSide(contigList, f);
    if (clipRightSide(contigList, f)) ret = true;
    return ret;
}
void SemialignedEndsClipper::clip(const ContigList &contigList, BamTemplate &t)
{
    for (unsigned k = 0; k < t.getFragmentCount(); ++k)
    {
        FragmentMetadata &fragment = t.getFragmentMetadata(k);
        if (clip(contigList, fragment) && 2 == t.getFragmentCount())
        {
            FragmentMetadata &mate = t.getMateFragmentMetadata(fragment);
            if (!mate.isAligned()) { mate.position = fragment.position; break; }
        }
    }
}

// matchSelector/OverlappingEndsClipper.cpp:46-183
void OverlappingEndsClipper::clip(const ContigList &contigList, BamTemplate &t)
{
    if (2 != t.getFragmentCount()) return;
    FragmentMetadata &r1 = t.getFragmentMetadata(0);
    FragmentMetadata &r2 = t.getFragmentMetadata(1);
    if (!r1.isAligned() || !r2.isAligned() || r1.gapCount || r2.gapCount) return;
    // :62 compares r1.contigId with itself: chimeric pairs are NOT skipped
    if (r1.isReverse() == r2.isReverse()) return;
    FragmentMetadata &left = r1.position < r2.position ? r1 : r2;
    FragmentMetadata &right = r1.position <= r2.position ? r2 : r1;
    if (left.isReverse()) return;
    const long overlapLength = left.position + left.getObservedLength() - right.position;
    if (0 >= overlapLength) return;
    unsigned leftEndSoftClip = 0;
    unsigned leftEndOffset = left.getReadLength();
    unsigned leftLastIdx = left.cigarOffset + left.cigarLength - 1;
    std::pair<unsigned, CigarOp> leftLastOp = cigarDecode(left.cigarBuffer->at(leftLastIdx));
    if (SOFT_CLIP == leftLastOp.second)
    {
        leftEndOffset -= leftLastOp.first; leftEndSoftClip = leftLastOp.first;
        --leftLastIdx; leftLastOp = cigarDecode(left.cigarBuffer->at(leftLastIdx));
    }
    if (ALIGN != leftLastOp.second) throw std::logic_error("Apart from soft-clipping, CIGAR must end with align operations.");
    if (overlapLength >= long(leftLastOp.first)) return;
    unsigned rightStartOffset = 0;
    unsigned rightFirstIdx = right.cigarOffset;
    std::pair<unsigned, CigarOp> rightFirstOp = cigarDecode(right.cigarBuffer->at(rightFirstIdx));
    if (SOFT_CLIP == rightFirstOp.second) { rightStartOffset += rightFirstOp.first; ++rightFirstIdx; rightFirstOp = cigarDecode(right.cigarBuffer->at(rightFirstIdx)); }
    if (ALIGN != rightFirstOp.second) throw std::logic_error("Apart from soft-clipping, CIGAR must begin with align operations.");
    if (overlapLength >= long(rightFirstOp.first)) return;
    int diff = 0;
    {
        const std::vector<char> &lq = left.getRead().forwardQuality; const std::vector<char> &rq = right.getRead().reverseQuality;
        for (long i = 0; i < overlapLength; ++i) diff += int(lq[leftEndOffset - overlapLength + i]) - int(rq[rightStartOffset + i]);
    }
    if (0 < diff)
    {
        const char *reference = contigList.at(right.contigId).forward.data() + right.position;
        const std::vector<uint32_t> tail(right.cigarBuffer->begin() + rightFirstIdx + 1, right.cigarBuffer->begin() + right.cigarOffset + right.cigarLength);
        right.cigarOffset = unsigned(cigarBuffer.size());
        cigarBuffer.push_back(cigarEncode(unsigned(rightStartOffset + overlapLength), SOFT_CLIP));
        cigarBuffer.push_back(cigarEncode(unsigned(rightFirstOp.first - overlapLength), ALIGN));
        cigarBuffer.insert(cigarBuffer.end(), tail.begin(), tail.end());
        right.incrementClipLeft((unsigned short)overlapLength);
        right.observedLength -= unsigned(overlapLength);
        const std::vector<char> &rs = right.getRead().reverseSequence;
        int ed = 0; for (long i = 0; i < overlapLength; ++i) ed += (rs[rightStartOffset + i] != reference[i]);
        right.editDistance -= ed;
        right.cigarBuffer = &cigarBuffer;
        right.cigarLength = unsigned(cigarBuffer.size()) - right.cigarOffset;
    }
    else
    {
        const char *reference = contigList.at(left.contigId).forward.data() + left.position + left.getObservedLength() - overlapLength;
        const std::vector<uint32_t> head(left.cigarBuffer->begin() + left.cigarOffset, left.cigarBuffer->begin() + leftLastIdx);
        left.cigarOffset = unsigned(cigarBuffer.size());
        cigarBuffer.insert(cigarBuffer.end(), head.begin(), head.end());
        cigarBuffer.push_back(cigarEncode(unsigned(leftLastOp.first - overlapLength), ALIGN));
        cigarBuffer.push_back(cigarEncode(unsigned(leftEndSoftClip + overlapLength), SOFT_CLIP));
        left.incrementClipRight((unsigned short)overlapLength);
        left.observedLength -= unsigned(overlapLength);
        const std::vector<char> &ls = left.getRead().forwardSequence;
        int ed = 0; for (long i = 0; i < overlapLength; ++i) ed += (ls[leftEndOffset - overlapLength + i] != reference[i]);
        left.editDistance -= ed;
        left.cigarBuffer = &cigarBuffer;
        left.cigarLength = unsigned(cigarBuffer.size()) - left.cigarOffset;
    }
}

// ---------------------------------------------------------------- parity record
// include/io/Fragment.hh:101-188 (+ getTlen :217-246) and include/build/FragmentAccessorBamAdapter.hh:250-265
static int getTlen(const FragmentMetadata &fragment, const FragmentMetadata &mate)
{
    if (!fragment.isAligned() || !mate.isAligned()) return 0;
    const ReferencePosition fb = fragment.getBeginReferencePosition(), fe = fragment.getEndReferencePosition();
    const ReferencePosition mb = mate.getBeginReferencePosition(), me = mate.getEndReferencePosition();
    const unsigned long distance = std::max(fe, me).getLocation() - std::min(fb, mb).getLocation();
    const bool firstRead = 0 == fragment.getReadIndex();
    const long ret = fb < mb ? long(distance) : (mb < fb || !firstRead) ? -long(distance) : long(distance);
    return int(ret);
}

FragmentRecord makeFragmentRecord(const BamTemplate &t, const FragmentMetadata &f, const FragmentMetadata *mate, int dodgyAlignmentScore)
{
    FragmentRecord r; memset(&r, 0, sizeof(r));
    const uint16_t DODGY = 0xffff;
    if (mate)
    {
        r.bamTlen = getTlen(f, *mate);
        r.fStrandPosition = (f.isAligned() ? f.getFStrandReferencePosition() : mate->getFStrandReferencePosition()).value;
        r.templateAlignmentScore = uint16_t(t.isProperPair() ? t.getAlignmentScore() : f.getAlignmentScore());
        r.mateFStrandPosition = (mate->isAligned() ? mate->getFStrandReferencePosition() : f.getFStrandReferencePosition()).value;
        r.flags = 1u | (unsigned(!f.isAligned()) << 1) | (unsigned(!mate->isAligned()) << 2) | (unsigned(f.isReverse()) << 3) | (unsigned(mate->isReverse()) << 4) |
                  (unsigned(0 == f.getReadIndex()) << 5) | (unsigned(1 == f.getReadIndex()) << 6) | (unsigned(!f.cluster->pf) << 7) | (unsigned(t.isProperPair()) << 8);
    }
    else
    {
        r.bamTlen = 0;
        r.fStrandPosition = f.getFStrandReferencePosition().value;
        r.templateAlignmentScore = uint16_t(f.getAlignmentScore());
        r.mateFStrandPosition = ReferencePosition(ReferencePosition::NoMatch).value;
        r.flags = 0u | (unsigned(!f.isAligned()) << 1) | (1u << 2) | (unsigned(f.isReverse()) << 3) | (1u << 5) | (1u << 6) | (unsigned(!f.cluster->pf) << 7);
    }
    r.observedLength = f.getObservedLength();
    r.lowClipped = f.lowClipped; r.highClipped = f.highClipped;
    r.alignmentScore = uint16_t(f.getAlignmentScore());
    r.readLength = uint16_t(f.getReadLength());
    r.cigarLength = uint16_t(f.cigarLength);
    r.gapCount = uint16_t(f.getGapCount());
    r.editDistance = uint16_t(f.getEditDistance());
    r.tile = f.cluster->tile; r.clusterId = uint32_t(f.cluster->id);
    // bits 16-31: BamTemplate::getAlignmentScore as 16 bits (0xffff = unknown, -1U): the one input of io::getTemplateDuplicateRank
    // (Fragment.hh:66-71) that is neither in the FragmentHeader fields above nor in the reads
    r.reserved = uint32_t(uint16_t(t.getAlignmentScore())) << 16;
    const unsigned forced = unsigned(dodgyAlignmentScore) & 0xff;
    if (r.flags & (1u << 8))
        r.mapq = (DODGY == r.templateAlignmentScore) ? forced : std::min<unsigned>(60U, std::max(r.alignmentScore, r.templateAlignmentScore));
    else
        r.mapq = (DODGY == r.alignmentScore) ? forced : std::min<unsigned>(60U, r.alignmentScore);
    return r;
}

// ---------------------------------------------------------------- MatchSelector
MatchSelector::MatchSelector(const Params &p, const ContigList &c) : params(p), contigs(c), templateBuilder(p), tld(p.mateDriftRange) {}

static const Match *findNextCluster(const Match *it, const Match *end)
{
    if (it == end) return end;
    const uint64_t cluster = SeedId(it->seedId).getCluster();
    while ((++it != end) && cluster == SeedId(it->seedId).getCluster()) { }
    return it;
}

// MatchSelector.cpp:188-256: no quality trimming, no gaps, pf clusters whose first match is not NoMatch, until stable
TemplateLengthStatistics MatchSelector::determineTemplateLength(const Match *mb, const Match *me, const uint8_t *bcl, unsigned tile)
{
    tld.clear();
    if (2 != params.reads.size()) return tld.stats;
    Cluster cluster;
    const unsigned clusterLength = params.clusterLength();
    for (const Match *matchBegin = mb, *matchEnd = findNextCluster(mb, me); me != matchBegin && !tld.stats.stable;
         matchBegin = matchEnd, matchEnd = findNextCluster(matchBegin, me))
    {
        const uint64_t clusterId = SeedId(matchBegin->seedId).getCluster();
        if (!ReferencePosition::fromValue(matchBegin->location).isNoMatch())
        {
            cluster.init(params.reads, bcl + clusterId * clusterLength, tile, clusterId, true);
            templateBuilder.buildFragments(contigs, params.reads, params.seeds, matchBegin, matchEnd, cluster, false);
            tld.addTemplate(templateBuilder.fragmentBuilder.fragments);
        }
    }
    if (!tld.isStable()) tld.finalize();
    return tld.stats;
}

// MatchSelector.cpp:258-368 (single thread; pf == true for FASTQ input)
void MatchSelector::selectTile(const Match *mb, const Match *me, const uint8_t *bcl, unsigned tile, const TemplateLengthStatistics &tls,
                               std::vector<FragmentRecord> &records, std::vector<uint32_t> &cigarPool)
{
    const RestOfGenomeCorrection rog(contigs, params.reads);
    Cluster cluster;
    const unsigned clusterLength = params.clusterLength();
    BamTemplate &bamTemplate = templateBuilder.bamTemplate;
    for (const Match *matchBegin = mb; me != matchBegin;)
    {
        const Match *matchEnd = findNextCluster(matchBegin, me);
        const uint64_t clusterId = SeedId(matchBegin->seedId).getCluster();
        cluster.init(params.reads, bcl + clusterId * clusterLength, tile, clusterId, true);
        trimLowQualityEnds(cluster, params.baseQualityCutoff);
        bool store = false;
        if (ReferencePosition::fromValue(matchBegin->location).isNoMatch())
        {
            bamTemplate.initialize(params.reads, cluster);
            store = params.keepUnaligned;
        }
        else if (templateBuilder.buildFragments(contigs, params.reads, params.seeds, matchBegin, matchEnd, cluster, true))
        {
            if (templateBuilder.buildTemplate(contigs, rog, params.reads, cluster, tls, params.mapqThreshold) || params.keepUnaligned)
            {
                if (params.clipSemialigned) { semialignedClipper.reset(); semialignedClipper.clip(contigs, bamTemplate); }
                if (params.clipOverlapping) { overlappingClipper.reset(); overlappingClipper.clip(contigs, bamTemplate); }
                store = true;
            }
        }
        else
        {
            bamTemplate.initialize(params.reads, cluster);
            store = params.keepUnaligned;
        }
        if (store)
        {
            for (unsigned i = 0; i < bamTemplate.getFragmentCount(); ++i)
            {
                const FragmentMetadata &f = bamTemplate.getFragmentMetadata(i);
                const FragmentMetadata *mate = 2 == bamTemplate.getFragmentCount() ? &bamTemplate.getFragmentMetadata(1 - i) : 0;
                FragmentRecord r = makeFragmentRecord(bamTemplate, f, mate, params.dodgyAlignmentScore);
                r.cigarOffset = uint32_t(cigarPool.size());
                if (f.isAligned()) cigarPool.insert(cigarPool.end(), f.cigarBuffer->begin() + f.cigarOffset, f.cigarBuffer->begin() + f.cigarOffset + f.cigarLength);
                records.push_back(r);
            }
        }
        matchBegin = matchEnd;
    }
}

} // namespace oracle
