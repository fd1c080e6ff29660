// The .bai file next to sorted.bam (host code, no GPU work): what bam::BamIndexer / bam::BamIndex build while build::Build saves its bins
// (lib/bam/BamIndexer.cpp:43-126 BamIndexPart, :129-472 BamIndex; include/bam/BamIndexer.hh:44-55 for the constants).
//
// The reference indexes one bin of its output at a time: the bin's records give chunks and linear-index entries in offsets of the bin's
// uncompressed bytes (BamIndexPart), and these are turned into virtual file offsets with the BGZF blocks the bin was compressed to
// (BamIndex::processIndexPart / resolveOffset).  A part here is the same thing: a run of records of the uncompressed stream that was
// compressed on its own, starting at a block boundary.  isaac-align (host/isaac_align.cpp) makes every contig a part, as the BAM stage
// makes every contig a bin, and the unaligned records the last part.
#include "../../include/isaac_gpu.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace
{
thread_local std::string g_indexError;
int indexFail(int code, const std::string &what) { g_indexError = what; return code; }

const uint32_t MAX_BIN = 37450;                 // BAM_MAX_BIN
const uint64_t UNSET = ~uint64_t(0);

uint32_t le32(const uint8_t *p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; }
uint32_t reg2bin(uint32_t beg, uint32_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

// where the BGZF blocks of a part begin, in compressed and in uncompressed bytes of the part
struct BlockTable
{
    std::vector<uint64_t> compressed, uncompressed;      // one entry per block + the totals at the end
    bool build(const uint8_t *bgzf, uint64_t n)
    {
        uint64_t at = 0, raw = 0;
        while (at < n)
        {
            if (at + 18 > n || 0x1f != bgzf[at] || 0x8b != bgzf[at + 1] || 8 != bgzf[at + 2] || 4 != bgzf[at + 3] || 'B' != bgzf[at + 12] || 'C' != bgzf[at + 13]) return false;
            const uint64_t size = uint64_t(bgzf[at + 16] | bgzf[at + 17] << 8) + 1;
            if (at + size > n) return false;
            compressed.push_back(at); uncompressed.push_back(raw);
            raw += le32(bgzf + at + size - 4);
            at += size;
        }
        compressed.push_back(at); uncompressed.push_back(raw);
        return true;
    }
    // BamIndex::resolveOffset: the block that holds uncompressed byte `u` of the part (a position on a block boundary belongs to the block
    // that starts there; the end of the part is offset 0 of whatever follows it)
    uint64_t resolve(uint64_t u, uint64_t partFileOffset) const
    {
        const size_t k = size_t(std::upper_bound(uncompressed.begin(), uncompressed.end(), u) - uncompressed.begin()) - 1;   // the last block (or the end) that starts at or before u
        return ((partFileOffset + compressed[k]) << 16) | (u - uncompressed[k]);
    }
};

struct Chunk { uint64_t begin, end; uint32_t bin, refId; };

struct ContigIndex
{
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t> > > bins;
    std::vector<uint64_t> linear;
    uint64_t mapped = 0, unmapped = 0;
};

void put(std::vector<uint8_t> &out, const void *p, size_t n) { const uint8_t *b = static_cast<const uint8_t *>(p); out.insert(out.end(), b, b + n); }
void put32(std::vector<uint8_t> &out, uint32_t v) { put(out, &v, 4); }
void put64(std::vector<uint8_t> &out, uint64_t v) { put(out, &v, 8); }

// BamIndex::outputBaiChromosomeIndex
void writeContig(std::vector<uint8_t> &out, const ContigIndex &c)
{
    if (!c.bins.empty() || c.mapped || c.unmapped)
    {
        put32(out, uint32_t(c.bins.size()) + 1);
        uint64_t first = 0, last = 0;
        for (const auto &bin : c.bins)
        {
            put32(out, bin.first); put32(out, uint32_t(bin.second.size()));
            for (const auto &chunk : bin.second) { put64(out, chunk.first); put64(out, chunk.second); }
            if (first > bin.second.front().first || !first) first = bin.second.front().first;
            if (last < bin.second.back().second || !last) last = bin.second.back().second;
        }
        put32(out, MAX_BIN); put32(out, 2); put64(out, first); put64(out, last); put64(out, c.mapped); put64(out, c.unmapped);   // samtools' bin of counts
    }
    else put32(out, 0);
    put32(out, uint32_t(c.linear.size()));
    for (const uint64_t v : c.linear) put64(out, v);
}

// the state BamIndex keeps between the bins it is given
struct Indexer
{
    std::vector<uint8_t> out; ContigIndex contig; uint32_t nContigs = 0, written = 0; uint64_t noCoordinates = 0, fileOffset = 0; uint32_t parts = 0;
};

// what BamIndexPart::processFragment reads of a record
struct RecordFacts { uint64_t at, length; int32_t refId, pos; uint32_t flag, seqLength, observed; };

// one part: BamIndexPart::processFragment over its records (next(facts) hands them over in file order; false at the end, a negative return of
// error() afterwards means a malformed part), BamIndex::processIndexPart with its blocks
template <typename NextF>
int addPartFrom(Indexer &x, uint64_t recordsBytes, const uint8_t *bgzf, uint64_t bgzfBytes, NextF next)
{
    const uint32_t p = x.parts++;
    if (!bgzfBytes) return 0;
    if (!bgzf) return indexFail(ISAAC_GPU_EINVAL, "a part without its bytes");
    BlockTable blocks;
    if (!blocks.build(bgzf, bgzfBytes)) return indexFail(ISAAC_GPU_EFORMAT, "part " + std::to_string(p) + ": not a run of BGZF blocks");
    if (blocks.uncompressed.back() != recordsBytes) return indexFail(ISAAC_GPU_EFORMAT, "part " + std::to_string(p) + ": the BGZF blocks do not hold records_bytes bytes");
    std::vector<Chunk> chunks;
    std::vector<uint64_t> linear;
    uint64_t mapped = 0, unmapped = 0;
    RecordFacts f;
    int rc = 0;
    while (next(f, rc))
    {
        const int32_t refId = f.refId, pos = f.pos;
        const uint64_t at = f.at;
        if (pos >= 0)
        {
            if (refId < 0 || uint32_t(refId) >= x.nContigs) return indexFail(ISAAC_GPU_EFORMAT, "record with a position on contig " + std::to_string(refId));
            if (uint32_t(pos) >= 512u * 1024 * 1024) return indexFail(ISAAC_GPU_EINVAL, "alignment position greater than the maximum allowed by BAM index: " + std::to_string(pos));
            const uint32_t bin = reg2bin(uint32_t(pos), uint32_t(pos) + f.seqLength);    // "samtools is doing it this way"
            const uint64_t end = at + f.length;
            // addToBinIndexChunks: the chunk grows while the bin stays; a record of the bin before last may still reach back to it
            if (!chunks.empty() && bin == chunks.back().bin && uint32_t(refId) == chunks.back().refId) chunks.back().end = end;
            else if (chunks.size() >= 2 && bin == chunks[chunks.size() - 2].bin && uint32_t(refId) == chunks[chunks.size() - 2].refId && chunks[chunks.size() - 2].end + 32768 /* BAM_MIN_CHUNK_GAP */ > end)
                chunks[chunks.size() - 2].end = end;
            else chunks.push_back(Chunk{ at, end, bin, uint32_t(refId) });
            // addToLinearIndex: the first record to reach a 16 kb window claims it, windows skipped on the way repeat the one before
            const uint32_t windows[2] = { uint32_t(pos) >> 14, f.observed ? (uint32_t(pos) + f.observed - 1) >> 14 : uint32_t(pos) >> 14 };
            for (const uint32_t w : windows)
                if (linear.size() <= w) { const uint64_t fill = linear.empty() ? UNSET : linear.back(); linear.resize(w + 1, fill); linear[w] = at; }
        }
        ++((f.flag & 4) ? unmapped : mapped);
    }
    if (rc) return indexFail(ISAAC_GPU_EFORMAT, "part " + std::to_string(p) + ": truncated record");
    if (!chunks.empty())
    {
        const uint32_t refId = chunks.front().refId;
        if (refId < x.written) return indexFail(ISAAC_GPU_EINVAL, "the parts are not in contig order");
        while (x.written < refId) { writeContig(x.out, x.contig); x.contig = ContigIndex(); ++x.written; }
        for (const Chunk &c : chunks)
        {
            const uint64_t begin = blocks.resolve(c.begin, x.fileOffset), end = blocks.resolve(c.end, x.fileOffset);
            std::vector<std::pair<uint64_t, uint64_t> > &bin = x.contig.bins[c.bin];
            if (!bin.empty() && (bin.back().second >> 16) == (begin >> 16)) bin.back().second = end;        // "small chunks reduction"
            else bin.push_back(std::make_pair(begin, end));
        }
        if (x.contig.linear.size() < linear.size()) x.contig.linear.resize(linear.size(), 0);
        for (size_t w = 0; w < linear.size(); ++w)
        {
            if (UNSET == linear[w]) continue;
            const uint64_t v = blocks.resolve(linear[w], x.fileOffset);
            if (v < x.contig.linear[w] || !x.contig.linear[w]) x.contig.linear[w] = v;
        }
        x.contig.mapped += mapped; x.contig.unmapped += unmapped;
    }
    else x.noCoordinates += unmapped;
    x.fileOffset += bgzfBytes;
    return 0;
}
// the records themselves as the source
int addPart(Indexer &x, const uint8_t *r, uint64_t recordsBytes, const uint8_t *bgzf, uint64_t bgzfBytes)
{
    if (bgzfBytes && recordsBytes && !r) return indexFail(ISAAC_GPU_EINVAL, "a part without its bytes");
    uint64_t at = 0;
    return addPartFrom(x, recordsBytes, bgzf, bgzfBytes, [&](RecordFacts &f, int &rc)
    {
        if (at >= recordsBytes) return false;
        if (at + 36 > recordsBytes) { rc = 1; return false; }
        const uint8_t *b = r + at;
        const uint64_t length = uint64_t(le32(b)) + 4;
        if (at + length > recordsBytes) { rc = 1; return false; }
        const uint32_t nameLength = b[12], flagNc = le32(b + 16), nCigar = flagNc & 0xffff;
        f.at = at; f.length = length; f.refId = int32_t(le32(b + 4)); f.pos = int32_t(le32(b + 8)); f.flag = flagNc >> 16; f.seqLength = le32(b + 20);
        f.observed = 0;                                                     // the reference bases the alignment covers
        for (uint32_t k = 0; k < nCigar; ++k) { const uint32_t w = le32(b + 36 + nameLength + 4 * k), op = w & 15; if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) f.observed += w >> 4; }
        at += length;
        return true;
    });
}
// what isaac_gpu_bam_records left about every record it wrote (isaac_bam_options::index_entries_dev) as the source
int addPartEntries(Indexer &x, const isaac_bam_index_entry *entries, uint64_t n, uint64_t recordsBytes, const uint8_t *bgzf, uint64_t bgzfBytes)
{
    if (n && !entries) return indexFail(ISAAC_GPU_EINVAL, "a part without its entries");
    uint64_t k = 0;
    return addPartFrom(x, recordsBytes, bgzf, bgzfBytes, [&](RecordFacts &f, int &rc)
    {
        if (k >= n) return false;
        const isaac_bam_index_entry &e = entries[k++];
        if (e.offset + e.bytes > recordsBytes) { rc = 1; return false; }
        f.at = e.offset; f.length = e.bytes; f.refId = e.ref_id; f.pos = e.pos; f.flag = e.flag; f.seqLength = e.seq_length; f.observed = e.observed;
        return true;
    });
}
// BamIndex::outputIndexFile
int finish(Indexer &x, uint8_t *baiOut, uint64_t capacity, uint64_t *nBytesOut)
{
    std::vector<uint8_t> out(x.out);
    ContigIndex contig = x.contig;
    for (uint32_t written = x.written; written < x.nContigs; ++written) { writeContig(out, contig); contig = ContigIndex(); }
    put64(out, x.noCoordinates);
    *nBytesOut = out.size();
    if (out.size() > capacity) return indexFail(ISAAC_GPU_ECAPACITY, "bai_out is too small");
    if (!baiOut) return indexFail(ISAAC_GPU_EINVAL, "bai_out is required");
    std::memcpy(baiOut, out.data(), out.size());
    return 0;
}
void start(Indexer &x, uint32_t nContigs, uint64_t headerBgzfBytes) { x.nContigs = nContigs; x.fileOffset = headerBgzfBytes; put(x.out, "BAI\1", 4); put32(x.out, nContigs); }
} // namespace

struct isaac_bam_indexer { Indexer x; };

extern "C" {

const char *isaac_gpu_bam_index_last_error(void) { return g_indexError.c_str(); }

int isaac_gpu_bam_index(const uint8_t *records, const isaac_bam_index_part *parts, uint32_t nParts, uint32_t nContigs, uint64_t headerBgzfBytes,
                        uint8_t *baiOut, uint64_t capacity, uint64_t *nBytesOut)
{
    if (nBytesOut) *nBytesOut = 0;
    if ((nParts && !parts) || !nBytesOut) return indexFail(ISAAC_GPU_EINVAL, "parts and n_bytes_out are required");
    Indexer x; start(x, nContigs, headerBgzfBytes);
    for (uint32_t p = 0; p < nParts; ++p)
        if (const int rc = addPart(x, records ? records + parts[p].records_offset : nullptr, parts[p].records_bytes, parts[p].bgzf_host, parts[p].bgzf_bytes)) return rc;
    return finish(x, baiOut, capacity, nBytesOut);
}

isaac_bam_indexer *isaac_gpu_bam_indexer_create(uint32_t nContigs, uint64_t headerBgzfBytes)
{
    isaac_bam_indexer *i = new (std::nothrow) isaac_bam_indexer;
    if (i) start(i->x, nContigs, headerBgzfBytes);
    return i;
}
int isaac_gpu_bam_indexer_add(isaac_bam_indexer *i, const uint8_t *records, uint64_t recordsBytes, const uint8_t *bgzf, uint64_t bgzfBytes)
{
    if (!i) return indexFail(ISAAC_GPU_EINVAL, "indexer is required");
    return addPart(i->x, records, recordsBytes, bgzf, bgzfBytes);
}
int isaac_gpu_bam_indexer_add_entries(isaac_bam_indexer *i, const isaac_bam_index_entry *entries, uint64_t nEntries, uint64_t recordsBytes, const uint8_t *bgzf, uint64_t bgzfBytes)
{
    if (!i) return indexFail(ISAAC_GPU_EINVAL, "indexer is required");
    return addPartEntries(i->x, entries, nEntries, recordsBytes, bgzf, bgzfBytes);
}
int isaac_gpu_bam_indexer_finish(isaac_bam_indexer *i, uint8_t *baiOut, uint64_t capacity, uint64_t *nBytesOut)
{
    if (!i || !nBytesOut) return indexFail(ISAAC_GPU_EINVAL, "indexer and n_bytes_out are required");
    return finish(i->x, baiOut, capacity, nBytesOut);
}
void isaac_gpu_bam_indexer_destroy(isaac_bam_indexer *i) { delete i; }

} // extern "C"
