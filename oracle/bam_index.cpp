// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  CPU restatement of the BAM index the reference writes next to sorted.bam:
//   bam::BamIndexPart   lib/bam/BamIndexer.cpp:30-126   what one bin's records contribute, in offsets of the bin's uncompressed bytes
//   bam::BamIndex       lib/bam/BamIndexer.cpp:129-472  resolving those against the bin's BGZF blocks, merging per contig, the .bai layout
//   constants           include/bam/BamIndexer.hh:44-55,646-647
// The classes keep the reference's shape (a part per bin, fed fragment by fragment; an index fed part by part with the part's compressed
// bytes), so that the product's one-pass formulation (isaac_aligner_amd/csrc/bam_index.cpp) is checked against the original control flow.
// Parity unpinned by reference vectors: the reference has no unit test or fixture for its indexer; tests/test_cli.py checks the result
// against the BAI semantics independently (every record is found through its bin's chunks and the linear index).
#include "oracle.hpp"

#include <cstring>
#include <stdexcept>

namespace oracle
{
namespace
{

const uint32_t BAM_MAX_CONTIG_LENGTH = 512 * 1024 * 1024, BAM_MAX_BIN = 37450, BAM_FUNMAP = 4, BAM_MIN_CHUNK_GAP = 32768;
typedef uint64_t UnresolvedOffset;

unsigned reg2binIndex(int beg, int end)
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

struct UnresolvedBinIndexChunk { UnresolvedOffset startPos, endPos; uint32_t bin, refId; };

struct BamIndexPart
{
    UnresolvedOffset localUncompressedOffset = 0;
    std::vector<UnresolvedBinIndexChunk> chunks;
    std::vector<UnresolvedOffset> linearIndex;
    uint64_t bamStatsMapped = 0, bamStatsNmapped = 0;

    // :44-77
    void processFragment(const BamIndexRecord &alignment)
    {
        if (alignment.pos >= 0)
        {
            const uint32_t observedLength = alignment.observedLength;
            const uint32_t bin = reg2binIndex(alignment.pos, alignment.pos + alignment.seqLen);    // seqLen, not observedLength: "samtools is doing it this way"
            addToBinIndexChunks(localUncompressedOffset, localUncompressedOffset + alignment.serializedLength, bin, uint32_t(alignment.refId));
            addToLinearIndex(uint32_t(alignment.pos), localUncompressedOffset);
            if (observedLength > 0) addToLinearIndex(uint32_t(alignment.pos) + observedLength - 1, localUncompressedOffset);
        }
        if (alignment.flag & BAM_FUNMAP) ++bamStatsNmapped; else ++bamStatsMapped;
        localUncompressedOffset += alignment.serializedLength;
    }
    // :79-101
    void addToBinIndexChunks(UnresolvedOffset virtualOffset, UnresolvedOffset virtualEndOffset, uint32_t bin, uint32_t refId)
    {
        if (bin >= BAM_MAX_BIN) throw std::runtime_error("Invalid bin number in uncompressed BAM");
        if (!chunks.empty() && bin == chunks.back().bin && refId == chunks.back().refId) chunks.back().endPos = virtualEndOffset;
        else if (chunks.size() >= 2 && bin == chunks[chunks.size() - 2].bin && refId == chunks[chunks.size() - 2].refId && (chunks[chunks.size() - 2].endPos + BAM_MIN_CHUNK_GAP) > virtualEndOffset)
            chunks[chunks.size() - 2].endPos = virtualEndOffset;                                   // chunk reduction around the boundary of two adjacent bins
        else { const UnresolvedBinIndexChunk c = { virtualOffset, virtualEndOffset, bin, refId }; chunks.push_back(c); }
    }
    // :103-119
    void addToLinearIndex(uint32_t pos, UnresolvedOffset virtualOffset)
    {
        if (pos >= BAM_MAX_CONTIG_LENGTH) throw std::runtime_error("Alignment position greater than the maximum allowed by BAM index: " + std::to_string(pos));
        const uint32_t linearBin = pos >> 14;
        if (linearIndex.size() <= linearBin)
        {
            const UnresolvedOffset lastValue = linearIndex.empty() ? 0xFFFFFFFFFFFFFFFFull : linearIndex.back();
            while (linearIndex.size() <= linearBin) linearIndex.push_back(lastValue);
            linearIndex[linearBin] = virtualOffset;
        }
    }
};

struct VirtualOffset
{
    uint64_t val = 0;
    void set(uint64_t cOffset, uint32_t uOffset) { val = (cOffset << 16) | uOffset; }
    uint64_t compressedOffset() const { return val >> 16; }
};
typedef std::pair<VirtualOffset, VirtualOffset> VirtualOffsetPair;

class BamIndex
{
public:
    BamIndex(uint32_t bamRefCount, uint32_t bamHeaderCompressedLength, std::vector<char> &bai) :
        bamRefCount_(bamRefCount), lastProcessedRefId_(0xFFFFFFFF), bai_(bai), binIndex_(BAM_MAX_BIN), binIndexEmpty_(true), positionInBam_(bamHeaderCompressedLength)
    {
        write("BAI\1", 4); write(&bamRefCount_, 4);                                                // outputBaiHeader
    }
    // :294-339
    void processIndexPart(const BamIndexPart &part, const std::vector<char> &bgzfBuffer)
    {
        if (bgzfBuffer.empty()) return;
        if (!part.chunks.empty())
        {
            const uint32_t refId = part.chunks[0].refId;
            while (lastProcessedRefId_ != refId)
            {
                if (lastProcessedRefId_ == 0xFFFFFFFF) { clearStructures(); lastProcessedRefId_ = 0; }
                else
                {
                    if (lastProcessedRefId_ >= refId) throw std::runtime_error("Bam indexer tries to process more chromosomes than was declared in Bam header");
                    outputBaiChromosomeIndex(); lastProcessedRefId_++;
                }
            }
            resetBgzfParsing();
            for (const UnresolvedBinIndexChunk &chunk : part.chunks) addToBinIndex(chunk, bgzfBuffer);            // mergeBinIndex
            mergeLinearIndex(part.linearIndex, bgzfBuffer);
            bamStatsMapped_ += part.bamStatsMapped; bamStatsNmapped_ += part.bamStatsNmapped;
        }
        else bamStatsGlobalNoCoordinates_ += part.bamStatsNmapped;                                                // block of unmapped reads
        positionInBam_ += bgzfBuffer.size();
    }
    // :189-203
    void outputIndexFile()
    {
        if (lastProcessedRefId_ == 0xFFFFFFFF) lastProcessedRefId_ = 0;
        while (lastProcessedRefId_ != bamRefCount_)
        {
            if (lastProcessedRefId_ >= bamRefCount_) throw std::runtime_error("Bam indexer processed more chromosomes than was declared in Bam header");
            outputBaiChromosomeIndex(); lastProcessedRefId_++;
        }
        write(&bamStatsGlobalNoCoordinates_, 8);                                                                  // outputBaiFooter
    }
private:
    void write(const void *p, size_t n) { const char *c = static_cast<const char *>(p); bai_.insert(bai_.end(), c, c + n); }
    // :214-281
    void outputBaiChromosomeIndex()
    {
        struct { uint32_t binNum, nClusters; uint64_t offBeg, offEnd, mapped, nmapped; } __attribute__((packed)) specialBin = { BAM_MAX_BIN, 2, 0, 0, bamStatsMapped_, bamStatsNmapped_ };
        uint32_t nBin = 0;
        if (!binIndexEmpty_) for (const std::vector<VirtualOffsetPair> &entry : binIndex_) nBin += !entry.empty();
        if (nBin > 0 || bamStatsMapped_ > 0 || bamStatsNmapped_ > 0)
        {
            ++nBin;                                                                                               // samtools' special bin
            write(&nBin, 4);
            uint32_t i = 0;
            for (const std::vector<VirtualOffsetPair> &entry : binIndex_)
            {
                if (!entry.empty())
                {
                    const uint32_t nChunk = uint32_t(entry.size());
                    write(&i, 4); write(&nChunk, 4); write(&entry[0], nChunk * 16);
                    if (specialBin.offBeg > entry[0].first.val || specialBin.offBeg == 0) specialBin.offBeg = entry[0].first.val;
                    if (specialBin.offEnd < entry[nChunk - 1].second.val || specialBin.offEnd == 0) specialBin.offEnd = entry[nChunk - 1].second.val;
                }
                ++i;
            }
            write(&specialBin, sizeof(specialBin));
        }
        else write(&nBin, 4);
        const uint32_t nIntv = uint32_t(linearIndex_.size());
        write(&nIntv, 4);
        if (!linearIndex_.empty()) write(&linearIndex_.front(), nIntv * 8);
        clearStructures();
    }
    // :384-400
    void mergeLinearIndex(const std::vector<UnresolvedOffset> &toMerge, const std::vector<char> &bgzfBuffer)
    {
        if (linearIndex_.size() < toMerge.size()) linearIndex_.resize(toMerge.size());
        for (unsigned i = 0; i < toMerge.size(); ++i)
            if (toMerge[i] != 0xFFFFFFFFFFFFFFFFull)
            {
                const VirtualOffset off = resolveOffset(toMerge[i], bgzfBuffer);
                if (off.val < linearIndex_[i].val || linearIndex_[i].val == 0) linearIndex_[i] = off;
            }
    }
    // :402-419
    void addToBinIndex(const UnresolvedBinIndexChunk &chunk, const std::vector<char> &bgzfBuffer)
    {
        const VirtualOffset start = resolveOffset(chunk.startPos, bgzfBuffer), end = resolveOffset(chunk.endPos, bgzfBuffer);
        if (!binIndex_[chunk.bin].empty() && binIndex_[chunk.bin].back().second.compressedOffset() == start.compressedOffset()) binIndex_[chunk.bin].back().second = end;   // small chunks reduction
        else binIndex_[chunk.bin].push_back(std::make_pair(start, end));
        binIndexEmpty_ = false;
    }
    void clearStructures()
    {
        bamStatsMapped_ = bamStatsNmapped_ = 0;
        if (!binIndexEmpty_) for (std::vector<VirtualOffsetPair> &entry : binIndex_) entry.clear();
        binIndexEmpty_ = true;
        linearIndex_.clear();
        resetBgzfParsing();
    }
    void resetBgzfParsing() { currentBgzfBlockCompressedPosition_ = currentBgzfBlockUncompressedPosition_ = 0; currentBgzfBlockCompressedSize_ = currentBgzfBlockUncompressedSize_ = 0; }
    // :436-468
    VirtualOffset resolveOffset(UnresolvedOffset unresolvedPos, const std::vector<char> &bgzfBuffer)
    {
        if (unresolvedPos < currentBgzfBlockUncompressedPosition_) resetBgzfParsing();
        while (unresolvedPos >= currentBgzfBlockUncompressedPosition_ + currentBgzfBlockUncompressedSize_)
        {
            currentBgzfBlockCompressedPosition_ += currentBgzfBlockCompressedSize_;
            currentBgzfBlockUncompressedPosition_ += currentBgzfBlockUncompressedSize_;
            if (currentBgzfBlockCompressedPosition_ == bgzfBuffer.size()) { currentBgzfBlockCompressedSize_ = 0; currentBgzfBlockUncompressedSize_ = 0; break; }
            const unsigned char *b = reinterpret_cast<const unsigned char *>(&bgzfBuffer[currentBgzfBlockCompressedPosition_]);
            if (currentBgzfBlockCompressedPosition_ + 13 >= bgzfBuffer.size() || b[0] != 0x1f || b[1] != 0x8b || b[2] != 8 || b[3] != 4 || b[12] != 0x42 || b[13] != 0x43)
                throw std::runtime_error("Error while parsing BGZF block during indexing");
            uint16_t compressedBlockSize; memcpy(&compressedBlockSize, b + 16, 2);
            uint32_t uncompressedBlockSize; memcpy(&uncompressedBlockSize, b + compressedBlockSize - 3, 4);
            currentBgzfBlockCompressedSize_ = uint64_t(compressedBlockSize) + 1;
            currentBgzfBlockUncompressedSize_ = uncompressedBlockSize;
        }
        VirtualOffset result;
        result.set(currentBgzfBlockCompressedPosition_ + positionInBam_, uint32_t(unresolvedPos - currentBgzfBlockUncompressedPosition_));
        return result;
    }

    uint32_t bamRefCount_, lastProcessedRefId_;
    std::vector<char> &bai_;
    std::vector<std::vector<VirtualOffsetPair> > binIndex_;
    bool binIndexEmpty_;
    std::vector<VirtualOffset> linearIndex_;
    uint64_t bamStatsMapped_ = 0, bamStatsNmapped_ = 0, bamStatsGlobalNoCoordinates_ = 0, positionInBam_;
    uint64_t currentBgzfBlockCompressedPosition_ = 0, currentBgzfBlockUncompressedPosition_ = 0, currentBgzfBlockCompressedSize_ = 0, currentBgzfBlockUncompressedSize_ = 0;
};

} // namespace

void bamIndex(const std::vector<BamIndexPartInput> &parts, uint32_t nContigs, uint32_t headerCompressedLength, std::vector<char> &bai)
{
    BamIndex index(nContigs, headerCompressedLength, bai);
    for (const BamIndexPartInput &in : parts)
    {
        BamIndexPart part;
        for (const BamIndexRecord &r : in.records) part.processFragment(r);
        index.processIndexPart(part, in.bgzf);
    }
    index.outputIndexFile();
}

} // namespace oracle
