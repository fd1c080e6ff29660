// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the Isaac seed-and-extend hot path, written from the
// behaviour of the reference sources (cited per function as file:line relative
// to /root/reference/src/c++).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, link or call anything in this
// directory.  The product path (isaac_aligner_amd/) never includes it.
//
// Pinning status: see oracle/README.md.  BandedSmithWaterman, SimpleIndelAligner,
// FragmentBuilder (from seed match lists; gapped/ungapped decision), SeedId, ClusterInfo,
// KmerGenerator, Permutate, NeighborsFinder::findNeighbors, TemplateLengthStatistics, the
// two end clippers, ShadowAligner::rescueShadow and TemplateBuilder::buildTemplate (MAPQ
// arithmetic, orphan rescue) are pinned by the reference's own cppunit known-answer
// vectors (tests/golden/).  The seed lookup proper (ClusterSeedGenerator, MatchFinder /
// ExactMaskMatcher) and the FASTQ reader have no reference vectors: "parity unpinned"
// for those rows.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <utility>
#include <limits>
#include <cmath>
#include <algorithm>
#include <stdexcept>

namespace oracle
{

// ---------------------------------------------------------------- formats
// include/alignment/Cigar.hh:52-70,156-168
enum CigarOp { ALIGN = 0, INSERT = 1, DELETE = 2, SKIP = 3, SOFT_CLIP = 4, HARD_CLIP = 5, PAD = 6, MATCH = 7, MISMATCH = 8, UNKNOWN = 9 };
inline uint32_t cigarEncode(unsigned len, CigarOp op) { return (len << 4) | unsigned(op); }
inline std::pair<unsigned, CigarOp> cigarDecode(uint32_t v)
{
    unsigned code = v & 0xF; if (code > 9) code = 9;
    return std::make_pair(v >> 4, CigarOp(code));
}
typedef std::vector<uint32_t> Cigar;
std::string cigarToString(const uint32_t *begin, const uint32_t *end);

// include/alignment/SeedId.hh:60-127
struct SeedId
{
    static constexpr unsigned REVERSE_WIDTH = 1, SEED_WIDTH = 8, CLUSTER_WIDTH = 31, BARCODE_WIDTH = 12, TILE_WIDTH = 12;
    static constexpr uint64_t REVERSE_MASK = 1, SEED_MASK = 0xFF, CLUSTER_MASK = 0x7FFFFFFFULL, BARCODE_MASK = 0xFFF, TILE_MASK = 0xFFF;
    static constexpr unsigned SEED_SHIFT = 1, CLUSTER_SHIFT = 9, BARCODE_SHIFT = 40, TILE_SHIFT = 52;
    uint64_t value;
    explicit SeedId(uint64_t v = 0) : value(v) {}
    // throws std::invalid_argument where the reference throws PreConditionException (SeedId.hh:95-109)
    SeedId(uint64_t tile, uint64_t barcode, uint64_t cluster, uint64_t seed, uint64_t reverse);
    uint64_t getTile() const { return (value >> TILE_SHIFT) & TILE_MASK; }
    uint64_t getBarcode() const { return (value >> BARCODE_SHIFT) & BARCODE_MASK; }
    uint64_t getCluster() const { return (value >> CLUSTER_SHIFT) & CLUSTER_MASK; }
    uint64_t getTileBarcode() const { return value >> BARCODE_SHIFT; }
    uint64_t getTileBarcodeCluster() const { return value >> CLUSTER_SHIFT; }
    uint64_t getSeed() const { return (value >> SEED_SHIFT) & SEED_MASK; }
    bool isNSeedId() const { return SEED_MASK == getSeed(); }
    bool isReverse() const { return value & 1; }
    void setNSeedId(bool lowestSeed) { value &= ~uint64_t(1); value |= (SEED_MASK << SEED_SHIFT) | uint64_t(!lowestSeed); }
    bool isLowestNSeedId() const { return isNSeedId() && !isReverse(); }
};

// include/reference/ReferencePosition.hh:51-188
struct ReferencePosition
{
    static constexpr unsigned CONTIG_ID_BITS = 23, POSITION_BITS = 40, NEIGHBORS_BITS = 1;
    static constexpr uint64_t MAX_CONTIG_ID = (~uint64_t(0)) >> (POSITION_BITS + NEIGHBORS_BITS);
    static constexpr uint64_t POSITION_MASK = ((~uint64_t(0)) >> (CONTIG_ID_BITS + NEIGHBORS_BITS)) << NEIGHBORS_BITS;
    static constexpr uint64_t POSITION_NEIGHBORS_MASK = (~uint64_t(0)) >> CONTIG_ID_BITS;
    uint64_t value;
    enum Special { TooManyMatch, NoMatch };
    explicit ReferencePosition(Special s) : value(((TooManyMatch == s ? 0 : MAX_CONTIG_ID) << POSITION_BITS) << NEIGHBORS_BITS) {}
    ReferencePosition() : value(0) {}
    ReferencePosition(uint64_t contigId, uint64_t position, bool neighbors = false)
        : value(((((contigId + 1) << POSITION_BITS) | position) << NEIGHBORS_BITS) | uint64_t(neighbors)) {}
    static ReferencePosition fromValue(uint64_t v) { ReferencePosition r; r.value = v; return r; }
    uint64_t getContigId() const { return (value >> (POSITION_BITS + NEIGHBORS_BITS)) - 1; }
    uint64_t getPosition() const { return (value & POSITION_MASK) >> NEIGHBORS_BITS; }
    uint64_t getLocation() const { return (value >> NEIGHBORS_BITS) - (uint64_t(1) << POSITION_BITS); }
    bool hasNeighbors() const { return value & 1; }
    bool isNoMatch() const { return value == ReferencePosition(NoMatch).value; }
    bool isTooManyMatch() const { return (value >> NEIGHBORS_BITS) == 0; }
    ReferencePosition &setNeighbors(bool n) { value = (value & ~uint64_t(1)) | uint64_t(n); return *this; }
    ReferencePosition translateContig(const std::vector<unsigned> &table) const
    {
        const unsigned contigValue = unsigned(value >> (POSITION_BITS + NEIGHBORS_BITS));
        if (contigValue)
            return fromValue(((uint64_t(table.at(contigValue - 1)) + 1) << (POSITION_BITS + NEIGHBORS_BITS)) | (value & POSITION_NEIGHBORS_MASK));
        return *this;
    }
    bool operator<(const ReferencePosition &p) const { return value < p.value; }
    bool operator==(const ReferencePosition &p) const { return value == p.value; }
    bool operator!=(const ReferencePosition &p) const { return value != p.value; }
};

// include/reference/ReferenceKmer.hh:37-54  (packed 16 bytes, as stored in mask files)
struct ReferenceKmer { uint64_t kmer; uint64_t position; };
// include/alignment/Match.hh:38-73
struct Match { uint64_t seedId; uint64_t location; };
// include/alignment/Seed.hh:42-93
struct Seed { uint64_t kmer; uint64_t seedId; };
// include/alignment/SeedMetadata.hh:43-101
struct SeedMetadata { unsigned offset; unsigned length; unsigned readIndex; unsigned index; };
// flowcell::ReadMetadata subset: length, index, offset of the read in the cluster, first/last cycle (1-based)
struct ReadMetadata { unsigned length; unsigned index; unsigned offset; unsigned firstCycle; unsigned lastCycle() const { return firstCycle + length - 1; } };

// lib/options/alignOptions/SeedDescriptorOption.cpp:90-151 ("auto"); returns first-pass-seed upper bound for the read
unsigned parseAutoSeedDescriptor(bool detectSimpleIndels, const ReadMetadata &read, unsigned seedLength, std::vector<SeedMetadata> &out);
// SeedDescriptorOption.cpp:209-245 for descriptor == "auto"; firstPassSeeds is reduced if reads are short
std::vector<SeedMetadata> autoSeeds(bool detectSimpleIndels, const std::vector<ReadMetadata> &reads, unsigned seedLength, unsigned &firstPassSeeds);
// lib/workflow/alignWorkflow/FindMatchesTransition.cpp:90-110
std::vector<std::vector<unsigned> > seedIndexListPerIteration(const std::vector<SeedMetadata> &seeds, unsigned nReads, unsigned firstPassSeeds);

// ---------------------------------------------------------------- parameters
// include/flowcell/SequencingAdapterMetadata.hh:33-73: sequence in the direction of the reference; reverse: the strand it is expected on; clipLength 0 = unbounded
// ("ACGT*" / "*ACGT" of --default-adapters, lib/options/AlignOptions.cpp:189-207)
struct SequencingAdapterMetadata
{
    std::string sequence; bool reverse; unsigned clipLength;
    bool isUnbounded() const { return !clipLength; }
};
struct Params
{
    std::vector<SequencingAdapterMetadata> adapters;      // --default-adapters of the flowcell (empty: nothing is clipped)
    int gapMatchScore = 0, gapMismatchScore = -3, gapOpenScore = -11, gapExtendScore = -4, minGapExtendScore = -20; // AlignOptions.cpp:55 "bwa"
    unsigned repeatThreshold = 10;          // AlignOptions.cpp:94
    unsigned gappedMismatchesMax = 5;       // AlignOptions.cpp:124
    unsigned semialignedGapLimit = 100;     // AlignOptions.cpp:131
    unsigned baseQualityCutoff = 25;        // AlignOptions.cpp:110
    bool ignoreNeighbors = false;
    bool clipSemialigned = true, clipOverlapping = true, scatterRepeats = false;
    int dodgyAlignmentScore = 0;            // TemplateBuilder::DodgyAlignmentScore; 255 = unknown, -1 = unaligned
    unsigned mapqThreshold = 0;
    bool pfOnly = true, keepUnaligned = true;
    int mateDriftRange = -1;
    unsigned firstPassSeeds = 2;
    unsigned seedLength = 32;
    std::vector<ReadMetadata> reads;
    std::vector<SeedMetadata> seeds;
    unsigned clusterLength() const { unsigned r = 0; for (size_t i = 0; i < reads.size(); ++i) r += reads[i].length; return r; }
    unsigned maxReadLength() const { unsigned r = 0; for (size_t i = 0; i < reads.size(); ++i) r = r > reads[i].length ? r : reads[i].length; return r; }
};
Params makeParams(unsigned nReads, unsigned len1, unsigned len2); // defaults + auto seeds

// ---------------------------------------------------------------- reference
struct Contig { unsigned index; std::string name; std::vector<char> forward; size_t getLength() const { return forward.size(); } };
typedef std::vector<Contig> ContigList;
size_t genomeLength(const ContigList &c); // lib/reference/Contig.cpp:30-38

struct SortedReference
{
    std::vector<ReferenceKmer> kmers;     // all masks concatenated == globally sorted by kmer (masks are the top bits)
    std::vector<unsigned> karyotype;      // contig index -> karyotype index
};
// lib/reference/ReferenceSorter.cpp:105-261 (+ neighbor flag semantics of NeighborsFinder.cpp:395-446, computed by
// brute-force Hamming search over the permutation blocks; small genomes only)
SortedReference buildSortedReference(const ContigList &contigs, unsigned seedLength, unsigned repeatThreshold /*1000*/, bool annotateNeighbors, unsigned neighborhoodWidth /*4*/, unsigned nThreads = 1);

// ---------------------------------------------------------------- banded smith-waterman
// lib/alignment/BandedSmithWaterman.cpp:36-54,84-462
struct BandedSmithWaterman
{
    static constexpr unsigned WIDEST_GAP_SIZE = 16, distanceCutoff = 7, mismatchesCutoff = 5;
    int matchScore, mismatchScore, gapOpenScore, gapExtendScore, maxReadLength;
    int16_t initialValue;
    mutable std::vector<uint8_t> T; // maxReadLength * 3 * 16 bytes
    BandedSmithWaterman(int match, int mismatch, int gapOpen, int gapExtend, int maxReadLength); // throws std::invalid_argument on overflow bound
    unsigned align(const char *queryBegin, const char *queryEnd, const char *dbBegin, const char *dbEnd, Cigar &cigar) const;
    // the rows of the band (flags into T, the last row's G / E / F out): lane by lane, or sixteen lanes at a time (AVX2); ORACLE_BSW_SCALAR=1 selects the former
    void rowsScalar(const char *queryBegin, size_t querySize, const char *dbBegin, int16_t *G, int16_t *E, int16_t *F) const;
    void rowsAvx2(const char *queryBegin, size_t querySize, const char *dbBegin, int16_t *G, int16_t *E, int16_t *F) const;
    static bool useAvx2();
};

void bswForceScalarRows(int on);     // this thread's alignments by the lane-by-lane rows (the check of the AVX2 rows against them)

// ---------------------------------------------------------------- quality
// lib/alignment/Quality.cpp:34-66, include/alignment/Quality.hh:51-112
struct Quality
{
    static const std::vector<double> &logMatchLookup();
    static const std::vector<double> &logMismatchLookup();
    static double getLogMatch(unsigned q) { return logMatchLookup()[q]; }
    static double getLogMismatch(unsigned q);
    static double getLogMismatchFast(unsigned q) { return logMismatchLookup()[q]; }
    static double restOfGenomeCorrection(unsigned genomeLength, unsigned readLength);
};
inline bool LP_EQUALS(double l, double r) { return 0.0000001 >= std::abs(l - r); }
inline bool LP_LESS(double l, double r) { return !LP_EQUALS(l, r) && l < r; }

// ---------------------------------------------------------------- reads
// include/alignment/Read.hh, lib/alignment/Read.cpp:32-73
struct Read
{
    unsigned index = 0;
    std::vector<char> forwardSequence, reverseSequence, forwardQuality, reverseQuality;
    unsigned endCyclesMasked = 0;
    const std::vector<char> &getStrandSequence(bool r) const { return r ? reverseSequence : forwardSequence; }
    const std::vector<char> &getStrandQuality(bool r) const { return r ? reverseQuality : forwardQuality; }
    unsigned getLength() const { return unsigned(forwardSequence.size()); }
    unsigned getBeginCyclesMasked() const { return 0; }
    unsigned getEndCyclesMasked() const { return endCyclesMasked; }
    void decodeBcl(const uint8_t *begin, const uint8_t *end, unsigned index);
};
// include/alignment/Cluster.hh, lib/alignment/Cluster.cpp:43-70
struct Cluster
{
    unsigned tile = 0; uint64_t id = 0; bool pf = true;
    const uint8_t *bcl = 0;
    Read reads[2]; unsigned nReads = 0;
    void init(const std::vector<ReadMetadata> &readMetadata, const uint8_t *bclData, unsigned tile, uint64_t id, bool pf);
    const Read &operator[](unsigned i) const { return reads[i]; }
    Read &operator[](unsigned i) { return reads[i]; }
};
void trimLowQualityEnd(Read &read, unsigned baseQualityCutoff);      // lib/alignment/Quality.cpp:72-105
void trimLowQualityEnds(Cluster &cluster, unsigned baseQualityCutoff); // Quality.cpp:107-120

// include/alignment/Alignment.hh:44-47
inline bool isMatch(char readBase, char referenceBase) { return readBase == 'n' || (readBase == referenceBase && referenceBase != 'N'); }

// ---------------------------------------------------------------- fragments
// include/alignment/FragmentMetadata.hh:48-483
struct FragmentMetadata
{
    const Cluster *cluster = 0;
    unsigned contigId = unsigned(ReferencePosition::MAX_CONTIG_ID);
    long position = 0;
    unsigned short lowClipped = 0, highClipped = 0;
    unsigned observedLength = 0;
    unsigned readIndex = 0;
    bool reverse = false;
    unsigned cigarOffset = 0, cigarLength = 0;
    const std::vector<uint32_t> *cigarBuffer = 0;
    unsigned mismatchCount = 0, matchesInARow = 0, gapCount = 0, editDistance = 0;
    std::vector<unsigned short> mismatchCycles;
    double logProbability = 0.0;
    int firstSeedIndex = -1;
    unsigned repeatSeedsCount = 0, uniqueSeedCount = 0;
    std::pair<unsigned, unsigned> nonUniqueSeedOffsets = std::make_pair(std::numeric_limits<unsigned>::max(), 0U);
    unsigned alignmentScore = -1U;
    unsigned smithWatermanScore = 0;

    FragmentMetadata() {}
    FragmentMetadata(const Cluster *c, const std::vector<uint32_t> *cb, unsigned ri) : cluster(c), readIndex(ri), cigarBuffer(cb) {}
    bool isReverse() const { return reverse; }
    unsigned getReadLength() const { return (*cluster)[readIndex].getLength(); }
    unsigned getReadIndex() const { return readIndex; }
    bool isAligned() const { return 0 != cigarLength; }
    unsigned getObservedLength() const { return isAligned() ? observedLength : 0; }
    unsigned getAlignmentScore() const { return alignmentScore; }
    void setAlignmentScore(unsigned as) { alignmentScore = as; }
    bool isNoMatch() const { return ReferencePosition::MAX_CONTIG_ID == contigId; }
    ReferencePosition getFStrandReferencePosition() const { return !isNoMatch() ? ReferencePosition(contigId, position) : ReferencePosition(ReferencePosition::NoMatch); }
    ReferencePosition getRStrandReferencePosition() const
    { return !isNoMatch() ? ReferencePosition(contigId, std::max(position + long(observedLength), 1L) - 1) : ReferencePosition(ReferencePosition::NoMatch); }
    ReferencePosition getBeginReferencePosition() const { return getFStrandReferencePosition(); }
    ReferencePosition getEndReferencePosition() const { return !isNoMatch() ? ReferencePosition(contigId, position + observedLength) : ReferencePosition(ReferencePosition::NoMatch); }
    const Read &getRead() const { return (*cluster)[readIndex]; }
    long getBeginClippedLength() const
    {
        if (cigarBuffer && cigarLength) { std::pair<unsigned, CigarOp> op = cigarDecode(cigarBuffer->at(cigarOffset)); if (SOFT_CLIP == op.second) return op.first; }
        return 0;
    }
    long getEndClippedLength() const
    {
        if (cigarBuffer && cigarLength) { std::pair<unsigned, CigarOp> op = cigarDecode(cigarBuffer->at(cigarOffset + cigarLength - 1)); if (SOFT_CLIP == op.second) return op.first; }
        return 0;
    }
    unsigned getMappedLength() const;
    long getUnclippedPosition() const { return position - getBeginClippedLength(); }
    unsigned getMismatchCount() const { return mismatchCount; }
    unsigned getGapCount() const { return gapCount; }
    unsigned getEditDistance() const { return editDistance; }
    void addMismatchCycle(unsigned cycle) { mismatchCycles.push_back((unsigned short)cycle); ++mismatchCount; }
    void setUnaligned() { cigarBuffer = 0; cigarLength = 0; alignmentScore = -1U; }
    void setNoMatch() { setUnaligned(); contigId = unsigned(ReferencePosition::MAX_CONTIG_ID); position = 0; }
    bool hasAlignmentScore() const { return -1U != alignmentScore; }
    void incrementClipLeft(unsigned short bases) { position += bases; if (reverse) highClipped += bases; else lowClipped += bases; }
    void incrementClipRight(unsigned short bases) { if (reverse) lowClipped += bases; else highClipped += bases; }
    unsigned short leftClipped() const { return reverse ? highClipped : lowClipped; }
    unsigned short rightClipped() const { return reverse ? lowClipped : highClipped; }
    unsigned short &leftClipped() { return reverse ? highClipped : lowClipped; }
    unsigned short &rightClipped() { return reverse ? lowClipped : highClipped; }
    void resetAlignment(Cigar &buffer)
    {
        position = getUnclippedPosition();
        cigarOffset = unsigned(buffer.size()); cigarLength = 0; cigarBuffer = &buffer; observedLength = 0;
        mismatchCycles.clear(); mismatchCount = 0; matchesInARow = 0; gapCount = 0; editDistance = 0;
        logProbability = 0.0; alignmentScore = -1U; smithWatermanScore = 0;
    }
    void resetClipping() { lowClipped = 0; highClipped = 0; }
    void consolidate(const FragmentMetadata &that)
    {
        uniqueSeedCount += that.uniqueSeedCount;
        nonUniqueSeedOffsets.first = std::min(nonUniqueSeedOffsets.first, that.nonUniqueSeedOffsets.first);
        nonUniqueSeedOffsets.second = std::max(nonUniqueSeedOffsets.second, that.nonUniqueSeedOffsets.second);
    }
    bool isWellAnchored() const
    { return uniqueSeedCount || (nonUniqueSeedOffsets.second > nonUniqueSeedOffsets.first && (nonUniqueSeedOffsets.second - nonUniqueSeedOffsets.first) >= 32 /*WEAK_SEED_LENGTH*/); }
    unsigned getContigId() const { return contigId; }
    long getPosition() const { return position; }
    bool operator<(const FragmentMetadata &f) const
    {
        return contigId < f.contigId || (contigId == f.contigId && (position < f.position ||
               (position == f.position && (reverse < f.reverse || (reverse == f.reverse && observedLength < f.observedLength)))));
    }
    bool operator==(const FragmentMetadata &t) const { return position == t.position && contigId == t.contigId && reverse == t.reverse && observedLength == t.observedLength; }
    bool operator!=(const FragmentMetadata &t) const { return !(*this == t); }
};
typedef std::vector<FragmentMetadata> FragmentMetadataList;

// lib/alignment/matchSelector/SequencingAdapter.cpp:30-141, include/alignment/matchSelector/SequencingAdapter.hh:37-69
struct SequencingAdapter
{
    static constexpr unsigned adapterMatchBasesMin = 5;
    static constexpr char UNINITIALIZED_POSITION = -1, NON_UNIQUE_KMER_POSITION = -2;
    SequencingAdapterMetadata metadata;
    std::vector<char> kmerPositions;
    explicit SequencingAdapter(const SequencingAdapterMetadata &m);
    // offsets into the strand sequence instead of iterators: [first, second), first == second: not found
    std::pair<long, long> getMatchRange(const char *sequence, long sequenceBegin, long sequenceEnd, long mismatchBase) const;
    bool isStrandCompatible(bool reverse) const { return !metadata.isUnbounded() || reverse == metadata.reverse; }
};
typedef std::vector<SequencingAdapter> SequencingAdapterList;

// lib/alignment/matchSelector/FragmentSequencingAdapterClipper.cpp:41-282, include/.../FragmentSequencingAdapterClipper.hh:34-86
struct FragmentSequencingAdapterClipper
{
    static constexpr unsigned TOO_GOOD_READ_MISMATCH_PERCENT = 40;
    const SequencingAdapterList &sequencingAdapters;
    struct Range { bool initialized = false, empty = true; long begin = 0, end = 0; } strandRange[2];
    explicit FragmentSequencingAdapterClipper(const SequencingAdapterList &a) : sequencingAdapters(a) {}
    void checkInitStrand(const FragmentMetadata &fragmentMetadata, const Contig &contig);
    void clip(const Contig &contig, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd) const;
    static bool decideWhichSideToClip(const Contig &contig, long contigPosition, const char *sequence, long sequenceLength, const Range &range, bool &clipBackwards);
};

// lib/alignment/fragmentBuilder/AlignerBase.cpp
struct AlignerBase
{
    unsigned normalizedMismatchScore, normalizedGapOpenScore, normalizedGapExtendScore, normalizedMaxGapExtendScore;
    AlignerBase(int match, int mismatch, int gapOpen, int gapExtend, int minGapExtend)
        : normalizedMismatchScore(match - mismatch), normalizedGapOpenScore(match - gapOpen),
          normalizedGapExtendScore(match - gapExtend), normalizedMaxGapExtendScore(-minGapExtend) {}
    static void clipReference(long referenceSize, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd);
    static void clipReadMasking(const Read &read, FragmentMetadata &fragment, const char *&sequenceBegin, const char *&sequenceEnd);
    unsigned updateFragmentCigar(const std::vector<ReadMetadata> &reads, const std::vector<char> &reference, FragmentMetadata &f,
                                 long strandPosition, const Cigar &cigarBuffer, unsigned cigarOffset) const;
};
// lib/alignment/fragmentBuilder/UngappedAligner.cpp:39-92
struct UngappedAligner : AlignerBase
{
    UngappedAligner(int a, int b, int c, int d, int e) : AlignerBase(a, b, c, d, e) {}
    unsigned alignUngapped(FragmentMetadata &f, Cigar &cigarBuffer, const std::vector<ReadMetadata> &reads, const FragmentSequencingAdapterClipper &adapterClipper, const Contig &contig) const;
};
// lib/alignment/fragmentBuilder/GappedAligner.cpp:51-82,167-249 (--avoid-smith-waterman 0 only)
struct GappedAligner : AlignerBase
{
    BandedSmithWaterman bsw;
    GappedAligner(int maxTotalReadLength, int a, int b, int c, int d, int e) : AlignerBase(a, b, c, d, e), bsw(a, b, -c, -d, maxTotalReadLength) {}
    unsigned alignGapped(FragmentMetadata &f, Cigar &cigarBuffer, const std::vector<ReadMetadata> &reads, const FragmentSequencingAdapterClipper &adapterClipper, const Contig &contig) const;
};
// lib/alignment/fragmentBuilder/SimpleIndelAligner.cpp
struct SimpleIndelAligner : AlignerBase
{
    static constexpr unsigned GAP_FLANK_BASES = 32, GAP_FLANK_MISMATCHES_MAX = 8;
    unsigned semialignedGapLimit;
    SimpleIndelAligner(int a, int b, int c, int d, int e, unsigned limit) : AlignerBase(a, b, c, d, e), semialignedGapLimit(limit) {}
    void alignSimpleIndels(Cigar &cigarBuffer, const ContigList &contigs, const std::vector<ReadMetadata> &reads,
                           const std::vector<SeedMetadata> &seeds, FragmentMetadataList &fragmentList) const;
    void alignSimpleDeletion(Cigar &cigarBuffer, FragmentMetadata &head, unsigned headSeedOffset, FragmentMetadata &tail,
                             unsigned tailSeedOffset, unsigned tailSeedLength, const ContigList &contigs, const std::vector<ReadMetadata> &reads) const;
    void alignSimpleInsertion(Cigar &cigarBuffer, FragmentMetadata &head, unsigned headSeedOffset, unsigned headSeedLength,
                              FragmentMetadata &tail, unsigned tailSeedOffset, unsigned tailSeedLength,
                              const ContigList &contigs, const std::vector<ReadMetadata> &reads) const;
};
// lib/alignment/FragmentBuilder.cpp
struct FragmentBuilder
{
    unsigned repeatThreshold, semialignedGapLimit, gappedMismatchesMax;
    std::vector<unsigned> seedMatchCounts;
    unsigned repeatSeedsCount;
    std::vector<FragmentMetadataList> fragments; // [2]
    Cigar cigarBuffer;
    UngappedAligner ungappedAligner; GappedAligner gappedAligner; SimpleIndelAligner simpleIndelAligner;
    SequencingAdapterList sequencingAdapters;      // (the reference passes the flowcell's list into build(); one flowcell here)
    FragmentBuilder(const Params &p);
    bool build(const ContigList &contigs, const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds,
               const Match *matchBegin, const Match *matchEnd, const Cluster &cluster, bool withGaps);
    void clear();
    void addMatch(const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds, const Match &m, const Cluster &cluster);
    void alignFragments(const ContigList &contigs, const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds, bool withGaps);
    static void consolidateDuplicateFragments(FragmentMetadataList &list, bool removeUnaligned);
};

// ---------------------------------------------------------------- oligo helpers
// include/oligo/KmerGenerator.hpp:39-131: successive N-free k-mers of an ASCII sequence (any non-ACGT byte restarts the k-mer).
// Used by ShadowAligner (7-mers of the shadow and of the rescue window, ShadowAligner.cpp:59,82).
template <typename T = unsigned>
struct KmerGenerator
{
    const char *current, *end; unsigned kmerLength; T mask; T kmer;
    static unsigned value(char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
    KmerGenerator(const char *b, const char *e, unsigned k) : current(b), end(e), kmerLength(k), mask(T(~((~T(0)) << (2 * k)))), kmer(0) { initialize(); }
    void initialize()
    {
        unsigned currentLength = 0;
        while ((current < end) && currentLength + 1 < kmerLength)
        {
            const unsigned v = value(*current);
            if (4 > v) { kmer <<= 2; kmer |= v; ++currentLength; } else { currentLength = 0; kmer = 0; }
            ++current;
        }
    }
    bool next(T &out, const char *&position)
    {
        while ((current < end) && (4 <= value(*current))) initialize();
        if (current < end)
        {
            kmer <<= 2; kmer |= value(*current); kmer &= mask; out = kmer; ++current; position = current - kmerLength;
            return true;
        }
        return false;
    }
};
// KmerGenerator.hpp:133-168
template <typename T> inline T getMaxKmer(unsigned kmerLength) { return T(~(~T(0) << 2 * kmerLength)); }
template <typename T> inline bool generateKmer(unsigned kmerLength, T &kmer, const char *current, const char *end)
{
    for (unsigned todo = kmerLength; todo; --todo, ++current)
    {
        if (current == end) return false;
        kmer <<= 2; kmer |= KmerGenerator<T>::value(*current);
    }
    kmer &= getMaxKmer<T>(kmerLength);
    return true;
}

// include/oligo/Permutate.hh:43-100, lib/oligo/Permutate.cpp:32-170: a k-mer as `count` blocks of `blockLength` bases (block 0 = the
// most significant); operator() takes a k-mer from the block order `from` to the block order `to`, reorder() from `to` back to
// the natural order
class Permutate
{
    unsigned blockLength_, count_;
    uint64_t order_, absoluteReverseOrder_;
    static uint64_t encode(const std::vector<unsigned> &from, const std::vector<unsigned> &to);
    template <typename KmerT> KmerT transform(KmerT kmer, uint64_t order) const
    {
        const unsigned blockBits = 2 * blockLength_;
        const KmerT blockMask = KmerT(~((~KmerT(0)) << blockBits));
        KmerT ret = 0;
        for (unsigned origin = 0; origin < count_; ++origin)
        {
            const unsigned target = unsigned((order >> ((count_ - origin - 1) * 4)) & 0xf);
            ret |= KmerT(((kmer >> ((count_ - origin - 1) * blockBits)) & blockMask) << ((count_ - target - 1) * blockBits));
        }
        return ret;
    }
public:
    Permutate(unsigned blockLength, const std::vector<unsigned> &from, const std::vector<unsigned> &to);
    template <typename KmerT> KmerT operator()(KmerT kmer) const { return transform(kmer, order_); }
    template <typename KmerT> KmerT reorder(KmerT kmer) const { return transform(kmer, absoluteReverseOrder_); }
};
// Permutate.cpp:94-145: for `errorCount` tolerated mismatches the k-mer is cut into 2 * errorCount blocks; every choice of
// errorCount blocks (ascending) becomes the prefix once.  Each Permutate leads from the previous order to the next one.
std::vector<Permutate> getPermutateList(unsigned kmerBases, unsigned errorCount);

// lib/reference/NeighborsFinder.cpp:286-383: in a list sorted by value, k-mers sharing the upper half are compared on the lower
// half; both ends of a pair 1..4 mismatches apart get the flag.
template <typename KmerT> struct AnnotatedKmer { KmerT value; bool hasNeighbors; bool operator<(const AnnotatedKmer &o) const { return value < o.value; } };
template <typename KmerT> void findNeighbors(std::vector<AnnotatedKmer<KmerT> > &kmerList, unsigned jobs, unsigned nThreads = 1);

// ---------------------------------------------------------------- seeds + matches
// include/alignment/matchFinder/TileClusterInfo.hh:65-143: two bytes per cluster; bit 0 of byte r = read r+1 complete, the six bits
// above it = one half of the 12-bit barcode index (all ones = no barcode assigned yet)
namespace matchFinder
{
class ClusterInfo
{
    uint8_t byte1_, byte2_;
    static const unsigned FOUND_MASK = 1, BARCODE_HALF_BITS = 6, BARCODE_HALF_MASK = ((1u << BARCODE_HALF_BITS) - 1) << 1;
public:
    static const unsigned MAX_BARCODE_VALUE = (1u << (2 * BARCODE_HALF_BITS)) - 1;
    ClusterInfo() : byte1_(0xff & ~FOUND_MASK), byte2_(0xff & ~FOUND_MASK) {}
    explicit ClusterInfo(bool markComplete) : byte1_(markComplete ? FOUND_MASK : 0), byte2_(markComplete ? FOUND_MASK : 0) {}
    unsigned getBarcodeIndex() const { return ((byte1_ & BARCODE_HALF_MASK) >> 1) | (((byte2_ & BARCODE_HALF_MASK) >> 1) << BARCODE_HALF_BITS); }
    bool isBarcodeSet() const { return MAX_BARCODE_VALUE != getBarcodeIndex(); }
    void setBarcodeIndex(unsigned barcodeIndex)
    {
        if (barcodeIndex >= MAX_BARCODE_VALUE) throw std::invalid_argument("Barcode does not fit in the allowed bit range");
        byte1_ = uint8_t((byte1_ & FOUND_MASK) | ((barcodeIndex << 1) & BARCODE_HALF_MASK));
        byte2_ = uint8_t((byte2_ & FOUND_MASK) | (((barcodeIndex >> BARCODE_HALF_BITS) << 1) & BARCODE_HALF_MASK));
    }
    bool isReadComplete(unsigned readIndex) const { return ((0 == readIndex) ? byte1_ : byte2_) & FOUND_MASK; }
    void markReadComplete(unsigned readIndex) { ((0 == readIndex) ? byte1_ : byte2_) |= FOUND_MASK; }
    void unmarkComplete() { byte1_ &= uint8_t(~FOUND_MASK); byte2_ &= uint8_t(~FOUND_MASK); }
    uint8_t byte(unsigned i) const { return i ? byte2_ : byte1_; }
};
}
// one tile of matchFinder::TileClusterInfo (:160-209)
typedef std::vector<matchFinder::ClusterInfo> ClusterInfo;
// lib/alignment/ClusterSeedGenerator.cpp:138-192 + SeedGeneratorBase.cpp:71-94 (sorted by (kmer, seedIndex))
void generateSeeds(const Params &p, const std::vector<unsigned> &seedIndexList, const uint8_t *bcl, unsigned nClusters,
                   unsigned tile, const ClusterInfo &complete, std::vector<Seed> &seeds);
// lib/alignment/MatchFinder.cpp:213-316 + matchFinder/ExactMaskMatcher.cpp:83-184 (merge join over the whole sorted reference)
void findMatchesExact(const Params &p, const SortedReference &ref, const std::vector<Seed> &sortedSeeds, bool closeRepeats, bool storeNoMatches,
                      ClusterInfo &complete, std::vector<Match> &matchesOut, std::vector<uint8_t> &contigHasMatches);
// FindMatchesTransition.cpp:391-427 for one tile: iteration 0 then iteration 1; matches sorted as SelectMatchesTransition.cpp:242-254.
// NoMatch records of reads already complete are never emitted (deterministic resolution of the benign race, Debug notes SURVEY §5).
void findTileMatches(const Params &p, const SortedReference &ref, const uint8_t *bcl, unsigned nClusters, unsigned tile,
                     std::vector<Match> &sortedMatches, std::vector<uint8_t> &contigHasMatches);
bool sortByTileBarcodeClusterLocation(const Match &l, const Match &r);
// the same on nThreads threads, each taking a range of the k-mer space as MatchFinder::matchMaskParallel hands out masks (MatchFinder.cpp:251-316)
void findTileMatchesParallel(const Params &p, const SortedReference &ref, const uint8_t *bcl, unsigned nClusters, unsigned tile, unsigned nThreads,
                             std::vector<Match> &sortedMatches, std::vector<uint8_t> &contigHasMatches);

// ---------------------------------------------------------------- template stage
// include/alignment/TemplateLengthStatistics.hh, lib/alignment/TemplateLengthStatistics.cpp
struct TemplateLengthStatistics
{
    enum AlignmentModel { FFp = 0, FRp = 1, RFp = 2, RRp = 3, FFm = 4, FRm = 5, RFm = 6, RRm = 7, InvalidAlignmentModel = 8 };
    enum CheckModelResult { Oversized = 0, Undersized = 1, Nominal = 2, NoMatch = 3 };
    static constexpr unsigned TEMPLATE_LENGTH_THRESHOLD = 50000;
    unsigned min = -1U, max = -1U, median = -1U, lowStdDev = -1U, highStdDev = -1U;
    AlignmentModel bestModels[2] = { InvalidAlignmentModel, InvalidAlignmentModel };
    bool stable = false;
    unsigned mateMin = -1U, mateMax = -1U;
    TemplateLengthStatistics() {}
    TemplateLengthStatistics(unsigned mn, unsigned mx, unsigned med, unsigned lo, unsigned hi, AlignmentModel m0, AlignmentModel m1, int mateDriftRange, bool stable_ = true)
        : min(mn), max(mx), lowStdDev(lo), highStdDev(hi), stable(stable_) { setMedian(med, mateDriftRange); bestModels[0] = m0; bestModels[1] = m1; }
    void clear() { *this = TemplateLengthStatistics(); }
    void setMin(unsigned v, int drift) { min = v; mateMin = -1 == drift ? min : median - drift; }
    void setMedian(unsigned v, int drift) { median = v; mateMin = -1 == drift ? min : median - drift; mateMax = -1 == drift ? max : median + drift; }
    void setMax(unsigned v, int drift) { max = v; mateMax = -1 == drift ? max : median + drift; }
    static AlignmentModel alignmentModel(const FragmentMetadata &f1, const FragmentMetadata &f2);
    static unsigned long getLength(const FragmentMetadata &f1, const FragmentMetadata &f2);
    static unsigned alignmentClass(AlignmentModel m) { return (m < 4) ? unsigned(m) : ((~unsigned(m)) & 3); }
    bool isCoherent() const { return bestModels[0] != bestModels[1] && alignmentClass(bestModels[0]) == alignmentClass(bestModels[1]); }
    CheckModelResult checkModel(const FragmentMetadata &f1, const FragmentMetadata &f2) const;
    bool matchModel(const FragmentMetadata &f1, const FragmentMetadata &f2) const;
    bool isValidModel(bool reverse, unsigned readIndex) const;
    bool firstFragment(bool reverse, unsigned readIndex) const;
    bool mateOrientation(unsigned readIndex, bool reverse) const;
    long mateMinPosition(unsigned readIndex, bool reverse, long position, const unsigned *readLengths) const;
    long mateMaxPosition(unsigned readIndex, bool reverse, long position, const unsigned *readLengths) const;
};
struct TemplateLengthDistribution
{
    static constexpr unsigned UPDATE_FREQUENCY = 10000;
    TemplateLengthStatistics stats; int mateDriftRange; std::vector<unsigned> lengthList;
    unsigned templateCount = 0, uniqueCount = 0, count = 0;
    std::vector<std::vector<unsigned> > histograms;
    explicit TemplateLengthDistribution(int drift) : mateDriftRange(drift), histograms(8) {}
    void clear();
    bool addTemplate(const std::vector<FragmentMetadataList> &fragments);
    bool finalize();
    void updateStatistics();
    bool isStable() const { return stats.stable; }
};

// include/alignment/RestOfGenomeCorrection.hh:44-88
struct RestOfGenomeCorrection
{
    double rogCorrectionList[2]; double rogCorrection;
    RestOfGenomeCorrection(const ContigList &contigs, const std::vector<ReadMetadata> &reads);
    double getReadRogCorrection(unsigned r) const { return rogCorrectionList[r]; }
    double getRogCorrection() const { return rogCorrection; }
};

// include/alignment/BamTemplate.hh, lib/alignment/BamTemplate.cpp
struct BamTemplate
{
    std::vector<FragmentMetadata> fragments; const std::vector<uint32_t> *cigarBuffer; unsigned alignmentScore = 0; bool properPair = false;
    explicit BamTemplate(const std::vector<uint32_t> &cb) : cigarBuffer(&cb) {}
    void initialize(const std::vector<ReadMetadata> &reads, const Cluster &cluster);
    unsigned getFragmentCount() const { return unsigned(fragments.size()); }
    FragmentMetadata &getFragmentMetadata(unsigned i) { return fragments[i]; }
    const FragmentMetadata &getFragmentMetadata(unsigned i) const { return fragments[i]; }
    FragmentMetadata &getMateFragmentMetadata(const FragmentMetadata &m) { return fragments.at(getFragmentCount() - 1 - m.getReadIndex()); }
    unsigned getAlignmentScore() const { return alignmentScore; }
    bool hasAlignmentScore() const { return -1U != alignmentScore; }
    void setAlignmentScore(unsigned a) { alignmentScore = a; }
    void setProperPair(bool p) { properPair = p; }
    bool isProperPair() const { return properPair; }
    bool filterLowQualityFragments(unsigned mapqThreshold);
};

// lib/alignment/ShadowAligner.cpp
struct ShadowAligner
{
    static constexpr unsigned shadowKmerLength = 7, shadowKmerCount = 1 << 14, candidatePositionsMax = 10000;
    unsigned gappedMismatchesMax; UngappedAligner ungappedAligner; GappedAligner gappedAligner;
    SequencingAdapterList sequencingAdapters;
    std::vector<short> shadowKmerPositions; Cigar shadowCigarBuffer; std::vector<long> shadowCandidatePositions;
    ShadowAligner(const Params &p);
    bool rescueShadow(const ContigList &contigs, const FragmentMetadata &orphan, FragmentMetadataList &shadowList, size_t shadowListCapacity,
                      const std::vector<ReadMetadata> &reads, const TemplateLengthStatistics &tls, long bestTemplateLength);
    void findShadowCandidatePositions(const char *refBegin, const char *refEnd, const std::vector<char> &shadowSequence);
};

// lib/alignment/TemplateBuilder.cpp
struct TemplateBuilder
{
    static constexpr unsigned TRACKED_REPEATS_MAX_ONE_READ = 1000, SKIP_ORPHAN_EDIT_DISTANCE = 3, DODGY_BUT_CLEAN_ALIGNMENT_SCORE = 10;
    static constexpr int DODGY_ALIGNMENT_SCORE_UNKNOWN = 255, DODGY_ALIGNMENT_SCORE_UNALIGNED = -1;
    typedef const FragmentMetadata *FragmentIterator;
    struct ShadowProbability
    {
        ReferencePosition pos; double logProbability; long observedLength;
        explicit ShadowProbability(const FragmentMetadata &s) : pos(s.getFStrandReferencePosition()), logProbability(s.logProbability), observedLength(s.getObservedLength())
        { pos.setNeighbors(s.isReverse()); }
        bool operator<(const ShadowProbability &t) const
        { return pos < t.pos || (pos == t.pos && (LP_LESS(logProbability, t.logProbability) || (LP_EQUALS(logProbability, t.logProbability) && observedLength < t.observedLength))); }
        bool operator==(const ShadowProbability &t) const { return pos == t.pos && LP_EQUALS(logProbability, t.logProbability) && observedLength == t.observedLength; }
    };
    struct PairProbability
    {
        ShadowProbability r1, r2;
        PairProbability(const FragmentMetadata &a, const FragmentMetadata &b) : r1(a), r2(b) {}
        double logProbability() const { return r1.logProbability + r2.logProbability; }
        bool operator<(const PairProbability &t) const
        {
            return r1.pos < t.r1.pos || (r1.pos == t.r1.pos && (r2.pos < t.r2.pos || (r2.pos == t.r2.pos &&
                   (LP_LESS(t.logProbability(), logProbability()) || (LP_EQUALS(logProbability(), t.logProbability()) &&
                   (r1.observedLength < t.r1.observedLength || (r1.observedLength == t.r1.observedLength && r2.observedLength < t.r2.observedLength)))))));
        }
        bool operator==(const PairProbability &t) const
        { return r1.pos == t.r1.pos && r2.pos == t.r2.pos && LP_EQUALS(logProbability(), t.logProbability()) && r1.observedLength == t.r1.observedLength && r2.observedLength == t.r2.observedLength; }
    };
    struct BestPairInfo
    {
        std::vector<FragmentIterator> bestPairFragments[2];
        double bestTemplateLogProbability; unsigned long bestTemplateScore; unsigned resolvedTemplateCount, bestPairEditDistance; double totalTemplateProbability;
        BestPairInfo() { clear(); }
        void clear();
        void init(FragmentIterator r1, FragmentIterator r2) { clear(); bestPairFragments[0].push_back(r1); bestPairFragments[1].push_back(r2); }
        long getBestTemplateLength() const;
    };
    bool scatterRepeats; int dodgyAlignmentScore;
    FragmentBuilder fragmentBuilder; BamTemplate bamTemplate; ShadowAligner shadowAligner;
    std::vector<uint32_t> cigarBuffer; FragmentMetadataList shadowList;
    std::vector<ShadowProbability> allShadowProbabilities[2]; std::vector<PairProbability> allPairProbabilities;
    FragmentMetadataList bestOrphanShadows[2];
    BestPairInfo bestCombinationPairInfo, bestRescuedPair;
    unsigned long rescueCalls = 0, rescueCandidates = 0; // work counters for the roofline formula

    explicit TemplateBuilder(const Params &p);
    bool buildFragments(const ContigList &contigs, const std::vector<ReadMetadata> &reads, const std::vector<SeedMetadata> &seeds,
                        const Match *mb, const Match *me, const Cluster &cluster, bool withGaps)
    { return fragmentBuilder.build(contigs, reads, seeds, mb, me, cluster, withGaps); }
    bool buildTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                       const Cluster &cluster, const TemplateLengthStatistics &tls, unsigned mapqThreshold);
    bool buildTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                       const std::vector<FragmentMetadataList> &fragments, const Cluster &cluster, const TemplateLengthStatistics &tls);
    FragmentIterator getBestFragment(const FragmentMetadataList &list) const;
    bool updateMappingScore(FragmentMetadata &fragment, const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls,
                            FragmentIterator listFragment, const FragmentMetadataList &list, bool forceWellAnchored) const;
    void locateBestPair(const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls, BestPairInfo &ret) const;
    bool buildPairedEndTemplate(const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls, const std::vector<FragmentMetadataList> &fragments, BestPairInfo &best);
    bool flagDodgyTemplate(FragmentMetadata &orphan, FragmentMetadata &shadow, BamTemplate &t) const;
    bool flagDodgyTemplate(FragmentMetadata &orphan, BamTemplate &t) const;
    bool rescueShadow(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                      const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls);
    bool buildDisjoinedTemplate(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                                const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls, const BestPairInfo &knownBestPair);
    bool scoreDisjoinedTemplate(const std::vector<FragmentMetadataList> &fragments, const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls,
                                const BestPairInfo &bestOrphans, const BestPairInfo &knownBestPair, unsigned bestOrphanIndex,
                                double totalShadowProbability, double totalOrphanProbability, const FragmentIterator bestDisjoinedFragments[2]);
    bool pickBestFragment(const RestOfGenomeCorrection &rog, const TemplateLengthStatistics &tls, const FragmentMetadataList &list);
    bool pickBestPair(const ContigList &contigs, const RestOfGenomeCorrection &rog, const std::vector<ReadMetadata> &reads,
                      const std::vector<FragmentMetadataList> &fragments, const TemplateLengthStatistics &tls);
    FragmentMetadata cloneWithCigar(const FragmentMetadata &right);
    static double sumUniqueShadowProbabilities(std::vector<ShadowProbability> &v);
    static double sumUniquePairProbabilities(std::vector<PairProbability> &v);
};

// lib/alignment/matchSelector/SemialignedEndsClipper.cpp, OverlappingEndsClipper.cpp
struct SemialignedEndsClipper
{
    static constexpr unsigned CONSECUTIVE_MATCHES_MIN = 5;
    Cigar cigarBuffer;
    void reset() { cigarBuffer.clear(); }
    bool clipLeftSide(const ContigList &contigs, FragmentMetadata &f);
    bool clipRightSide(const ContigList &contigs, FragmentMetadata &f);
    bool clip(const ContigList &contigs, FragmentMetadata &f);
    void clip(const ContigList &contigs, BamTemplate &t);
};
struct OverlappingEndsClipper
{
    Cigar cigarBuffer;
    void reset() { cigarBuffer.clear(); }
    void clip(const ContigList &contigs, BamTemplate &t);
};

// The parity record: the fields the reference persists per read in io::FragmentHeader
// (include/io/Fragment.hh:101-188) plus the BAM MAPQ derived from them
// (include/build/FragmentAccessorBamAdapter.hh:250-265) and its CIGAR.
struct FragmentRecord
{
    uint64_t fStrandPosition, mateFStrandPosition;
    int32_t bamTlen; uint32_t observedLength;
    uint16_t lowClipped, highClipped, alignmentScore, templateAlignmentScore;
    uint16_t readLength, cigarLength, gapCount, editDistance;
    uint32_t flags;      // bit0 paired,1 unmapped,2 mateUnmapped,3 reverse,4 mateReverse,5 first,6 second,7 failFilter,8 properPair
    uint32_t cigarOffset;
    uint32_t tile, clusterId;
    uint32_t mapq;
    uint32_t reserved;
};
FragmentRecord makeFragmentRecord(const BamTemplate &t, const FragmentMetadata &f, const FragmentMetadata *mate, int dodgyAlignmentScore);

// lib/alignment/MatchSelector.cpp:188-256 (TLS learning), :258-368 (per cluster), :370-443 (per tile)
struct MatchSelector
{
    Params params; const ContigList &contigs; TemplateBuilder templateBuilder; TemplateLengthDistribution tld;
    SemialignedEndsClipper semialignedClipper; OverlappingEndsClipper overlappingClipper;
    MatchSelector(const Params &p, const ContigList &contigs);
    TemplateLengthStatistics determineTemplateLength(const Match *mb, const Match *me, const uint8_t *bcl, unsigned tile);
    // processes every cluster of the tile in match order; appends 2 (paired) or 1 records per processed cluster
    void selectTile(const Match *mb, const Match *me, const uint8_t *bcl, unsigned tile, const TemplateLengthStatistics &tls,
                    std::vector<FragmentRecord> &records, std::vector<uint32_t> &cigarPool);
};

// fastq.cpp: FastqSeedSource's tile rule
unsigned fastqTileClustersMax(unsigned clustersAtATimeMax, unsigned seedCount);
void fastqDiscoverTiles(unsigned clustersLoaded, unsigned tileClustersMax, unsigned &currentTile, std::vector<std::pair<unsigned, unsigned> > &loadedTiles);

// bam.cpp: the BAM record stream of a set of tiles (build::Build: duplicates, gap realignment, order, serialisation) and the BAM header
struct BamTileInput { const uint8_t *bcl; const FragmentRecord *records; const uint32_t *cigars; uint64_t nRecords; std::string namePrefix;
                      std::string readGroup;                        // the barcode index of the tile's lane (FragmentAccessorBamAdapter::getFragmentRG); empty: BamOptions::readGroup
                      const TemplateLengthStatistics *tls = 0; };   // of the tile's barcode; NULL: BamOptions::tls
struct BamOptions { unsigned clusterLength, readOffset[2]; unsigned char forcedDodgyAlignmentScore; bool pessimisticMapQ; std::string readGroup, barcode;
                    bool markDuplicates = false, keepDuplicates = true;          // --mark-duplicates / --keep-duplicates (BinSorter.cpp:293-330)
                    bool realignGaps = false, realignVigorously = false, realignDodgy = false, clipSemialigned = true; const ContigList *contigs = 0; const TemplateLengthStatistics *tls = 0;   // --realign-gaps sample
                    // Every contig is a bin of its own unless it is cut: ascending ReferencePosition values at which a contig goes on into a further bin
                    // (alignment::BinMetadata stretches of --target-bin-size, include/alignment/matchSelector/BinIndexMap.hh:44-104).  build::Build works bin
                    // by bin: duplicates, gaps and realignment never look beyond the bin (lib/build/BinSorter.cpp:293-330,387-417)
                    std::vector<uint64_t> binCuts; };
// One end of a pair as the duplicate filter sees it: build::FStrandFragmentIndex / RStrandOrShadowFragmentIndex (include/build/FragmentIndex.hh:101-192)
// with the fields of the fragment its comparators look up (library = barcode or sample index, tile * 10^9 + cluster).
struct PairEndIndex
{
    uint64_t primary;        // fStrandPos_ (forward-strand ends) or anchor_.value_ (reverse-strand ends and shadows)
    uint64_t mateAnchor;     // mate_.anchor_.value_
    uint32_t mateInfo;       // mate_.info_.value_: shadow | reverse << 1 | storageBin << 2
    uint64_t library, duplicateClusterRank, globalClusterId;
    uint64_t tag;            // caller's handle
};
// DuplicatePairEndFilter::filterInput (include/build/DuplicatePairEndFilter.hh:45-107) with FDuplicateFilter / RSDuplicateFilter's less and
// equal_to (DuplicateFragmentIndexFiltering.hh:37-208; the two differ only in what `primary` is): sorts `ends` and flags every end
// that the reference would discard (or mark, with --keep-duplicates) as a duplicate of the best one before it
void filterDuplicates(std::vector<PairEndIndex> &ends, std::vector<char> &isDuplicate);
void bamRecords(const std::vector<BamTileInput> &tiles, const BamOptions &o, std::vector<char> &os, uint64_t &nRecords, uint64_t &unalignedOffset);
struct SqTags { std::string as, ur, m5; };    // SortedReferenceMetadata::Contig::bamSqAs_ / bamSqUr_ / bamM5_
void bamHeader(const std::string &commandLine, const std::string &description, const std::string &version, const std::vector<std::string> &headerLines,
               const std::vector<std::pair<std::string, uint32_t> > &refSeqs, std::vector<char> &os, const std::vector<SqTags> *tags = 0);

// ---- bam_index.cpp: the .bai file (bam::BamIndexPart / bam::BamIndex, lib/bam/BamIndexer.cpp)
// what BamIndexPart::processFragment reads from the adapter of one serialised record
struct BamIndexRecord { int refId, pos, seqLen; uint32_t observedLength, flag, serializedLength; };
// one bin as build::Build saves it: its records in file order and the BGZF bytes they were compressed to
struct BamIndexPartInput { std::vector<BamIndexRecord> records; std::vector<char> bgzf; };
void bamIndex(const std::vector<BamIndexPartInput> &parts, uint32_t nContigs, uint32_t headerCompressedLength, std::vector<char> &bai);

// ---- gap realigner (realign.cpp)
// gapRealigner::Gap (include/build/gapRealigner/Gap.hh:31-78): length > 0 deletion from the reference, < 0 insertion
struct RealignGap
{
    ReferencePosition pos; int length;
    RealignGap(ReferencePosition p, int l) : pos(p), length(l) {}
    unsigned getLength() const { return unsigned(length < 0 ? -length : length); }
    bool isInsertion() const { return 0 > length; }
    bool isDeletion() const { return 0 < length; }
    ReferencePosition endPos(bool fatInsertions) const;
    ReferencePosition deletionEndPos() const;
};
// build::RealignerGaps (include/build/GapRealigner.hh:37-128)
struct RealignerGaps
{
    std::vector<RealignGap> gapGroups, deletionEndGroups;
    void addGaps(ReferencePosition fStrandPosition, const uint32_t *cigarBegin, const uint32_t *cigarEnd);
    void finalizeGaps();
    void findGaps(ReferencePosition rangeBegin, ReferencePosition rangeEnd, std::vector<RealignGap> &foundGaps, size_t capacity) const;
};
// gapRealigner::OverlappingGapsFilter (OverlappingGapsFilter.hh:32-92)
struct OverlappingGapsFilter
{
    static const unsigned MAX_TRACKED_OVERLAPS = 30, MAX_TRACKED_DELETIONS = 30;
    unsigned maxChoice; std::vector<unsigned> overlappingGaps;
    explicit OverlappingGapsFilter(const std::vector<RealignGap> &gaps);
    unsigned findOverlaps(unsigned combination) const;
    unsigned next(unsigned combination) const;
private:
    void findOverlaps(const std::vector<RealignGap> &gaps);
};
// the fields of io::FragmentAccessor the realigner reads and writes; bases: the read as FragmentCollector stored it (forward-strand BCL bytes)
struct RealignFragment
{
    ReferencePosition fStrandPosition; uint64_t mateFStrandPosition; unsigned observedLength; uint16_t lowClipped, highClipped, alignmentScore, templateAlignmentScore, readLength, editDistance;
    uint32_t flags; const unsigned char *bases;
    unsigned leftClipped() const { return (flags & 8) ? highClipped : lowClipped; }
    unsigned rightClipped() const { return (flags & 8) ? lowClipped : highClipped; }
};
struct RealignIndex { ReferencePosition pos; const uint32_t *cigarBegin, *cigarEnd; };       // PackedFragmentBuffer::Index
struct GapRealigner
{
    static const unsigned MAX_GAPS_AT_A_TIME = 10;
    bool realignGapsVigorously, realignDodgyFragments; unsigned realignedGapsPerFragment, mismatchCost, gapOpenCost, gapExtendCost; bool clipSemialigned;
    const ContigList &reference;
    // realignedCigars must have room (capacity) for everything the call appends: indexes of other fragments point into it
    void realign(const RealignerGaps &realignerGaps, ReferencePosition binStartPos, ReferencePosition binEndPos, RealignIndex &index, RealignFragment &fragment,
                 std::vector<uint32_t> &realignedCigars, bool &changed) const;
    unsigned getAlignmentCost(const RealignFragment &fragment, const RealignIndex &index, unsigned &editDistance, int &mismatchesPercent) const;
    struct Impl;
};

void setLookupMode(int mode);   // seeds.cpp: 0 merge join, 1 bisection between seed k-mers (parallel lookup only)

} // namespace oracle
