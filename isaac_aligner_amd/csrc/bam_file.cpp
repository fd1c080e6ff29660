// Host side of the BAM output (no GPU work): the BAM header as bam::serializeHeader writes it (include/bam/Bam.hh:153-235) and the
// BGZF framing of bgzf::BgzfCompressor (include/bgzf/BgzfCompressor.hh:36-176, include/bgzf/Bgzf.hh:30-85; footer lib/bam/Bam.cpp:38-45).
// Blocks are independent gzip members, so they are deflated by a pool of threads (the reference compresses its bins in parallel the
// same way, lib/build/Build.cpp).  zlib is the only dependency.
#include "../../include/isaac_gpu.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace
{
thread_local std::string g_bamError;
int bamFail(int code, const std::string &what) { g_bamError = what; return code; }

// what a block takes at most: the reference lets a block consume 0xFFFF - 41 bytes so that even stored it fits 16 bits of BSIZE
const uint64_t BLOCK_INPUT = 0xFFFF - 41, BLOCK_HEADER = 18, BLOCK_FOOTER = 8;
const unsigned char EOF_BLOCK[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

void putLe32(std::string &s, uint32_t v) { for (int i = 0; i < 4; ++i) s.push_back(char((v >> (8 * i)) & 0xff)); }

// one BGZF block: gzip header with the BC extra field, raw deflate data, CRC32, ISIZE
int deflateBlock(const uint8_t *in, uint32_t n, int level, uint8_t *out /* 0x10000 bytes */, uint32_t *nOut)
{
    z_stream z; std::memset(&z, 0, sizeof(z));
    if (Z_OK != deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY)) return 1;
    z.next_in = const_cast<Bytef *>(in); z.avail_in = n; z.next_out = out + BLOCK_HEADER; z.avail_out = 0x10000 - BLOCK_HEADER - BLOCK_FOOTER;
    const int rc = deflate(&z, Z_FINISH);
    const uint32_t produced = uint32_t(z.total_out);
    deflateEnd(&z);
    // data that deflate cannot shrink may not fit the 16 bits of BSIZE at this level: such a block is stored instead (the reference
    // has no such case: it would write a BSIZE that wrapped)
    if (Z_STREAM_END != rc) return level ? deflateBlock(in, n, 0, out, nOut) : 1;
    const uint32_t total = uint32_t(BLOCK_HEADER + produced + BLOCK_FOOTER), bsize = total - 1;
    // boost::iostreams::gzip header: no name, no comment, mtime 0, XFL by level, OS unknown; FLG gets FEXTRA when the BC field is put in
    const unsigned char xfl = (9 == level) ? 2 : (1 == level ? 4 : 0);
    const unsigned char header[BLOCK_HEADER] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, xfl, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 0xff), (unsigned char)(bsize >> 8) };
    std::memcpy(out, header, BLOCK_HEADER);
    const uint32_t crc = uint32_t(crc32(crc32(0, Z_NULL, 0), in, n));
    uint8_t *f = out + BLOCK_HEADER + produced;
    for (int i = 0; i < 4; ++i) { f[i] = uint8_t(crc >> (8 * i)); f[4 + i] = uint8_t(n >> (8 * i)); }
    *nOut = total;
    return 0;
}
} // namespace

extern "C" {

const char *isaac_gpu_bam_last_error(void) { return g_bamError.c_str(); }

int isaac_gpu_bam_header(const char *commandLine, const char *description, const char *version, const char *const *headerLines, uint32_t nHeaderLines,
                         const char *const *contigNames, const uint32_t *contigLengths, const char *const *contigAs, const char *const *contigUr, const char *const *contigM5,
                         uint32_t nContigs, uint8_t *out, uint64_t capacity, uint64_t *nBytesOut)
{
    if (nBytesOut) *nBytesOut = 0;
    if ((nContigs && (!contigNames || !contigLengths)) || (nHeaderLines && !headerLines)) return bamFail(ISAAC_GPU_EINVAL, "contig names, lengths and header lines are required");
    std::string text = "@HD\tVN:1.0\tSO:coordinate\n@PG\tID:iSAAC\tPN:iSAAC\tCL:";
    text += commandLine ? commandLine : ""; text += "\t";
    if (description && *description) { text += "DS:"; text += description; text += "\t"; }
    text += "VN:"; text += version ? version : ""; text += "\n";
    for (uint32_t i = 0; i < nHeaderLines; ++i) { text += headerLines[i]; text += "\n"; }
    for (uint32_t i = 0; i < nContigs; ++i)
    {   // Bam.hh:196-213: AS, UR and M5 follow when the sorted-reference metadata carries them, in this order
        text += "@SQ\tSN:"; text += contigNames[i]; text += "\tLN:"; text += std::to_string(contigLengths[i]);
        if (contigAs && contigAs[i] && *contigAs[i]) { text += "\tAS:"; text += contigAs[i]; }
        if (contigUr && contigUr[i] && *contigUr[i]) { text += "\tUR:"; text += contigUr[i]; }
        if (contigM5 && contigM5[i] && *contigM5[i]) { text += "\tM5:"; text += contigM5[i]; }
        text += "\n";
    }
    std::string bin("BAM\1", 4);
    putLe32(bin, uint32_t(text.size())); bin += text; putLe32(bin, nContigs);
    for (uint32_t i = 0; i < nContigs; ++i)
    {
        const size_t l = std::strlen(contigNames[i]);
        putLe32(bin, uint32_t(l + 1)); bin.append(contigNames[i], l + 1); putLe32(bin, contigLengths[i]);
    }
    if (nBytesOut) *nBytesOut = bin.size();
    if (bin.size() > capacity) return bamFail(ISAAC_GPU_ECAPACITY, "header buffer is too small");
    std::memcpy(out, bin.data(), bin.size());
    return 0;
}

uint64_t isaac_gpu_bgzf_bound(uint64_t nBytes) { return ((nBytes + BLOCK_INPUT - 1) / BLOCK_INPUT) * 0x10000 + sizeof(EOF_BLOCK); }

int isaac_gpu_bgzf_compress(const uint8_t *data, uint64_t nBytes, int level, uint32_t nThreads, int eofBlock, uint8_t *out, uint64_t capacity, uint64_t *nBytesOut)
{
    if (nBytesOut) *nBytesOut = 0;
    if (nBytes && (!data || !out)) return bamFail(ISAAC_GPU_EINVAL, "data and out are required");
    if (level < 0 || level > 9) return bamFail(ISAAC_GPU_EINVAL, "gzip level 0..9");
    const uint64_t nBlocks = (nBytes + BLOCK_INPUT - 1) / BLOCK_INPUT;
    if (capacity < nBlocks * 0x10000 + (eofBlock ? sizeof(EOF_BLOCK) : 0)) return bamFail(ISAAC_GPU_ECAPACITY, "out must hold isaac_gpu_bgzf_bound(n_bytes) bytes");
    if (!nThreads) nThreads = 1;
    if (nThreads > nBlocks) nThreads = uint32_t(nBlocks ? nBlocks : 1);
    // every block is deflated into its own 64 KiB slot of `out`, then the slots are closed up in place
    std::vector<uint32_t> sizes(nBlocks, 0);
    std::atomic<uint64_t> next(0); std::atomic<int> failed(0);
    auto work = [&]()
    {
        for (uint64_t b = next++; b < nBlocks; b = next++)
        {
            const uint64_t from = b * BLOCK_INPUT, n = std::min<uint64_t>(BLOCK_INPUT, nBytes - from);
            if (deflateBlock(data + from, uint32_t(n), level, out + b * 0x10000, &sizes[b])) failed = 1;
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t t = 1; t < nThreads; ++t) threads.emplace_back(work);
    work();
    for (auto &t : threads) t.join();
    if (failed) return bamFail(ISAAC_GPU_ENOMEM, "deflate failed");
    uint64_t at = 0;
    for (uint64_t b = 0; b < nBlocks; ++b) { if (at != b * 0x10000) std::memmove(out + at, out + b * 0x10000, sizes[b]); at += sizes[b]; }
    if (eofBlock) { std::memcpy(out + at, EOF_BLOCK, sizeof(EOF_BLOCK)); at += sizeof(EOF_BLOCK); }
    if (nBytesOut) *nBytesOut = at;
    return 0;
}

} // extern "C"
