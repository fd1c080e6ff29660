// DEFLATE (RFC 1951) pieces shared by the device compressor (deflate_kernels.h), the host code that makes its Huffman tables
// (deflate_tables.cpp) and the CPU model of the encoder the tests use: symbol arithmetic and the bits of a token.
//
// The reference compresses every BAM block with zlib at --bam-gzip-level (default 1) on CPU threads (include/bgzf/BgzfCompressor.hh:36-176,
// wired into the BAM writer at lib/build/Build.cpp:181-254).  What has to be identical is the inflated stream, not the compressed bytes, so
// the device does not mimic zlib: every BGZF block is one dynamic-Huffman deflate block (BTYPE 10) whose two code tables are made once per
// call from the symbol statistics of a sample of the call's blocks and shared by all of them.
#pragma once
#include "types.h"

namespace isaac
{

static const u32 DEFLATE_LITLEN_SYMBOLS = 286, DEFLATE_DIST_SYMBOLS = 30, DEFLATE_MIN_MATCH = 4, DEFLATE_MAX_MATCH = 258, DEFLATE_WINDOW = 32768;
static const u32 DEFLATE_END_OF_BLOCK = 256;
static const u32 DEFLATE_HEADER_WORDS = 96;      // room for the dynamic block header: at most 3 + 14 + 19 * 3 + 316 * 7 bits

// code tables as the kernels read them: Huffman code (bit-reversed: deflate packs codes starting from their most significant bit into a
// stream that fills bytes from the least significant bit) in bits 0..15, its length in bits 16..20
struct DeflateTables
{
    u32 litLen[DEFLATE_LITLEN_SYMBOLS];
    u32 dist[DEFLATE_DIST_SYMBOLS];
    u32 header[DEFLATE_HEADER_WORDS];      // BFINAL = 1, BTYPE = 10, HLIT, HDIST, HCLEN, the code length code and the two tables' lengths in it
    u32 headerBits;
};

// length 3..258 -> code 0..28 (symbol 257 + code), number of extra bits and their value (RFC 1951 3.2.5)
ISAAC_HD u32 deflateLengthCode(u32 length, u32 &extraBits, u32 &extra)
{
    if (length == 258) { extraBits = 0; extra = 0; return 28; }
    const u32 l = length - 3;                       // 0..254
    if (l < 8) { extraBits = 0; extra = 0; return l; }
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 msb = 31u - u32(__clz(int(l)));
#else
    const u32 msb = 31u - u32(__builtin_clz(l));
#endif
    extraBits = msb - 2;                            // l in [8,16): 1 extra bit, [16,32): 2, ...
    extra = l & ((1u << extraBits) - 1);
    return 4 * extraBits + 4 + ((l >> extraBits) & 3);
}
// distance 1..32768 -> code 0..29
ISAAC_HD u32 deflateDistanceCode(u32 distance, u32 &extraBits, u32 &extra)
{
    const u32 d = distance - 1;                     // 0..32767
    if (d < 4) { extraBits = 0; extra = 0; return d; }
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 msb = 31u - u32(__clz(int(d)));
#else
    const u32 msb = 31u - u32(__builtin_clz(d));
#endif
    extraBits = msb - 1;
    extra = d & ((1u << extraBits) - 1);
    return 2 * msb + ((d >> extraBits) & 1);
}

// the bits of a literal / of a (length, distance) pair in stream order (first bit of the token = bit 0) and their number (at most 48)
ISAAC_HD u64 deflateLiteralBits(const u32 *litLen, u32 byte, u32 &nBits) { const u32 c = litLen[byte]; nBits = c >> 16; return c & 0xffffu; }
ISAAC_HD u64 deflateMatchBits(const u32 *litLen, const u32 *dist, u32 length, u32 distance, u32 &nBits)
{
    u32 lx, le, dx, de;
    const u32 lc = litLen[257 + deflateLengthCode(length, lx, le)], dc = dist[deflateDistanceCode(distance, dx, de)];
    u64 bits = lc & 0xffffu; u32 n = lc >> 16;
    bits |= u64(le) << n; n += lx;
    bits |= u64(dc & 0xffffu) << n; n += dc >> 16;
    bits |= u64(de) << n; n += dx;
    nBits = n;
    return bits;
}

} // namespace isaac
