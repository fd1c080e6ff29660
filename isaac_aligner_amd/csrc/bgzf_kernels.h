// BGZF framing without compression on the device (--bam-gzip-level 0): every block of the output is a gzip member with the BC extra
// field (include/bgzf/Bgzf.hh:30-85) around one stored deflate block, as bgzf::BgzfCompressor produces with gzip level 0
// (include/bgzf/BgzfCompressor.hh:36-176); the CRC-32 of a block is computed by the workgroup that copies it.
#pragma once
#include "types.h"

namespace isaac
{

static const u32 BGZF_BLOCK_INPUT = 0xFFFF - 41;        // BgzfCompressor::max_uncompressed_per_block_
static const u32 BGZF_STORED_OVERHEAD = 18 + 5 + 8;     // gzip header with the BC field, stored-block header, CRC32 + ISIZE
static const u32 CRC_POLY = 0xedb88320u;                // CRC-32 of RFC 1952, reflected

// a(x) * b(x) mod p(x) in the reflected representation (bit 31 = x^0)
ISAAC_HD u32 crcMultiply(u32 a, u32 b)
{
    u32 p = 0;
    for (u32 m = 1u << 31; m; m >>= 1)
    {
        if (a & m) p ^= b;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8 * nBytes) mod p; squares[k] = x^(2^k) mod p
ISAAC_HD u32 crcShiftOperator(const u32 *squares, u64 nBytes)
{
    u32 r = 1u << 31;                                   // x^0
    u64 bits = nBytes * 8;
    for (u32 k = 0; bits; ++k, bits >>= 1) if (bits & 1) r = crcMultiply(r, squares[k]);
    return r;
}
struct CrcConstants { u32 table[4][256]; u32 squares[40]; };   // table[k][b]: the remainder of byte b followed by k zero bytes (slicing by four)
inline void makeCrcConstants(CrcConstants &c)
{
    for (u32 i = 0; i < 256; ++i) { u32 v = i; for (u32 k = 0; k < 8; ++k) v = (v & 1) ? (v >> 1) ^ CRC_POLY : v >> 1; c.table[0][i] = v; }
    for (u32 k = 1; k < 4; ++k) for (u32 i = 0; i < 256; ++i) c.table[k][i] = (c.table[k - 1][i] >> 8) ^ c.table[0][c.table[k - 1][i] & 0xff];
    c.squares[0] = 1u << 30;                            // x^1
    for (u32 k = 1; k < 40; ++k) c.squares[k] = crcMultiply(c.squares[k - 1], c.squares[k - 1]);
}

#if defined(__HIPCC__)
// One workgroup per BGZF block, two phases.  (1) The block's input goes to LDS with the widest loads its alignment allows and from there
// to its place in the output as whole words on the output's own four-byte boundaries.  (2) Thread i computes the remainder of bytes [260 i, 260 i + 260) from a zero register (a stride
// of 65 words keeps the lanes on different LDS banks); the remainders are folded pairwise (left * x^(8 * length of right) + right), and the
// register value 0xffffffff the CRC starts from is carried over the whole length at the end.
// (The first version, every thread copying and checking its own piece straight from global memory, ran at 37 GB/s: 64 cache lines per load
// instruction and a dependent table look-up per byte behind each.)
static const u32 BGZF_THREADS = 256, BGZF_CHUNK = 260;
static_assert(BGZF_THREADS * BGZF_CHUNK >= BGZF_BLOCK_INPUT, "every byte of a block has a thread");
__global__ void __launch_bounds__(256) k_bgzf_store(const u8 *data, u64 nBytes, const CrcConstants *constants, u8 *out)
{
    __shared__ u32 table[4][256];
    __shared__ u32 squares[40];
    __shared__ u32 partial[BGZF_THREADS];
    __shared__ u32 lengths[BGZF_THREADS];
    __shared__ __attribute__((aligned(4))) u8 staged[BGZF_BLOCK_INPUT + 6];
    for (u32 k = 0; k < 4; ++k) table[k][threadIdx.x] = constants->table[k][threadIdx.x];
    if (threadIdx.x < 40) squares[threadIdx.x] = constants->squares[threadIdx.x];
    const u64 from = u64(blockIdx.x) * BGZF_BLOCK_INPUT;
    const u32 n = u32(nBytes - from < BGZF_BLOCK_INPUT ? nBytes - from : BGZF_BLOCK_INPUT);
    u8 *block = out + u64(blockIdx.x) * (BGZF_BLOCK_INPUT + BGZF_STORED_OVERHEAD);
    const u32 total = n + BGZF_STORED_OVERHEAD, bsize = total - 1;
    if (threadIdx.x < 23)
    {
        const u8 header[23] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, u8(bsize), u8(bsize >> 8),
                                1 /* BFINAL, stored */, u8(n), u8(n >> 8), u8(~n), u8(~n >> 8) };
        block[threadIdx.x] = header[threadIdx.x];
    }
    // in: the input starts on an even address (blocks are 65 494 bytes long), on a multiple of four for every other block
    const u8 *src = data + from;
    if (0 == (reinterpret_cast<u64>(src) & 3))
    {
        for (u32 i = 4 * threadIdx.x; i + 4 <= n; i += 4 * BGZF_THREADS) *reinterpret_cast<u32 *>(staged + i) = *reinterpret_cast<const u32 *>(src + i);
        if (threadIdx.x < (n & 3)) staged[(n & ~3u) + threadIdx.x] = src[(n & ~3u) + threadIdx.x];
    }
    else if (0 == (reinterpret_cast<u64>(src) & 1))
    {
        for (u32 i = 2 * threadIdx.x; i + 2 <= n; i += 2 * BGZF_THREADS) *reinterpret_cast<u16 *>(staged + i) = *reinterpret_cast<const u16 *>(src + i);
        if (0 == threadIdx.x && (n & 1)) staged[n - 1] = src[n - 1];
    }
    else for (u32 i = threadIdx.x; i < n; i += BGZF_THREADS) staged[i] = src[i];
    __syncthreads();
    // out: words on four-byte boundaries of the output, put together from the staged bytes; single bytes before the first and after the last
    {
        u8 *dst = block + 23;
        const u32 head = u32((4 - (reinterpret_cast<u64>(dst) & 3)) & 3) < n ? u32((4 - (reinterpret_cast<u64>(dst) & 3)) & 3) : n;
        if (threadIdx.x < head) dst[threadIdx.x] = staged[threadIdx.x];
        const u32 words = (n - head) / 4;
        for (u32 q = threadIdx.x; q < words; q += BGZF_THREADS)
        {
            const u8 *b = staged + head + 4 * q;
            *reinterpret_cast<u32 *>(dst + head + 4 * q) = u32(b[0]) | (u32(b[1]) << 8) | (u32(b[2]) << 16) | (u32(b[3]) << 24);
        }
        const u32 tail = n - head - 4 * words;
        if (threadIdx.x < tail) dst[head + 4 * words + threadIdx.x] = staged[head + 4 * words + threadIdx.x];
    }
    const u32 begin = threadIdx.x * BGZF_CHUNK, end = begin + BGZF_CHUNK < n ? begin + BGZF_CHUNK : n;
    u32 crc = 0;
    u32 i = begin;
    for (; i + 4 <= end; i += 4)
    {   // four bytes a step: the look-ups of a step do not depend on each other (begin is a multiple of four)
        const u32 w = crc ^ *reinterpret_cast<const u32 *>(staged + i);
        crc = table[3][w & 0xff] ^ table[2][(w >> 8) & 0xff] ^ table[1][(w >> 16) & 0xff] ^ table[0][w >> 24];
    }
    for (; i < end; ++i) crc = table[0][(crc ^ staged[i]) & 0xff] ^ (crc >> 8);
    partial[threadIdx.x] = crc; lengths[threadIdx.x] = begin < n ? end - begin : 0;
    __syncthreads();
    for (u32 step = 1; step < BGZF_THREADS; step <<= 1)
    {
        if (0 == (threadIdx.x & (2 * step - 1)) && lengths[threadIdx.x + step])
        {
            partial[threadIdx.x] = crcMultiply(partial[threadIdx.x], crcShiftOperator(squares, lengths[threadIdx.x + step])) ^ partial[threadIdx.x + step];
            lengths[threadIdx.x] += lengths[threadIdx.x + step];
        }
        __syncthreads();
    }
    if (0 == threadIdx.x)
    {
        const u32 value = ~(crcMultiply(0xffffffffu, crcShiftOperator(squares, n)) ^ partial[0]);
        u8 *footer = block + 23 + n;
        for (u32 k = 0; k < 4; ++k) { footer[k] = u8(value >> (8 * k)); footer[4 + k] = u8(n >> (8 * k)); }
    }
}
#endif

} // namespace isaac
