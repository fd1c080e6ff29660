// BGZF with compression on the device (--bam-gzip-level 1 and up): every block of the output is a gzip member with the BC extra field
// (include/bgzf/Bgzf.hh:30-85) around one dynamic-Huffman deflate block.  One wavefront per block:
//   * the block's input (DEFLATE_BLOCK_INPUT bytes) is staged in LDS;
//   * the wave walks it 64 positions a step: every lane hashes the four bytes at its position, takes the last earlier position with that
//     hash from a table in LDS (and leaves its own there), checks and extends the match four bytes at a time;
//   * the greedy parse of the step -- the first match at or beyond the end of the previous one wins, what it covers is skipped, the rest are
//     literals -- is settled with ballots, without a loop over positions;
//   * every lane codes its token with the call's Huffman tables (deflate_common.h), a prefix sum of the bit counts places the tokens, the
//     bits are OR-ed into a small ring of words in LDS, and whole words leave for the block's slot in global memory after every step;
//   * CRC-32 and the framing finish the member; a block that deflate cannot shrink is stored.
// The same walk with `histogram` set counts symbols instead of coding them: that is how the call's tables are made from a sample of
// its blocks (deflate_tables.cpp).  A second kernel closes the blocks up.
#pragma once
#include "bgzf_kernels.h"
#include "deflate_common.h"

namespace isaac
{

bool makeDeflateTables(const u64 *litLenCounts, const u64 *distCounts, DeflateTables &t);     // deflate_tables.cpp

// Input bytes per block.  bgzf::BgzfCompressor takes 0xFFFF - 41 (so that a block fits 16 bits of BSIZE even stored); any smaller size is as
// valid a BGZF file.  What decides here is LDS: a block's input is staged there, and a CU runs as many blocks side by side as fit.
// (MI355X, the bench's 2.5 GB record stream, one wavefront per block: 65 494 bytes and 8 192 hash slots, one block per CU: 7.9 GB/s, ratio 0.508;
// 4 096 slots, two per CU: 16.1 GB/s, 0.510; 32 768 bytes: 24-26 GB/s, 0.512; 24 576 bytes and 2 048 slots, four per CU: 35.3 GB/s, 0.514; 16 384 bytes:
// 42.1 GB/s, 0.518 -- zlib level 1 on the same stream: 0.520 at 1 GB/s on 256 host threads.  The walk is a chain of LDS round trips that a lone
// wavefront waits out one by one, so throughput follows the number of blocks in flight; the ratio hardly moves because on BAM records most of
// the gain is the Huffman coding of four-bit bases and a few quality values, not the matches.  profiles/r4_exp_deflate*.log)
// (Round 5, the 6.4 GB record stream of the bench: 24 576 bytes, four blocks per CU: 35.2 GB/s, ratio 0.5138; 20 480 bytes, five per CU: 42.4 GB/s, 0.5158 -- the default
// since; 16 384 bytes: 41.9 GB/s, 0.5179 (five per CU as well); 16 384 bytes and 1 024 hash slots, six per CU: 53.1 GB/s, 0.5211.  profiles/exp_r5_deflate_block.log)
#ifndef ISAAC_DEFLATE_BLOCK_INPUT
#define ISAAC_DEFLATE_BLOCK_INPUT 20480
#endif
static const u32 DEFLATE_BLOCK_INPUT = ISAAC_DEFLATE_BLOCK_INPUT;
static_assert(DEFLATE_BLOCK_INPUT <= BGZF_BLOCK_INPUT && 0 == (DEFLATE_BLOCK_INPUT & 1), "block input size");
#ifndef ISAAC_DEFLATE_HASH_BITS
#define ISAAC_DEFLATE_HASH_BITS 11
#endif
static const u32 DEFLATE_HASH_BITS = ISAAC_DEFLATE_HASH_BITS;
static const u32 DEFLATE_SLOT = 0x10000 + 64;      // a block's slot in the staging buffer: two bytes of padding (the deflate data then starts on a word), the member
static const u32 DEFLATE_SLOT_PAD = 2;
static const u32 DEFLATE_RING_WORDS = 256;         // bit ring: a step adds at most 64 x 48 bits = 96 words

#if defined(__HIPCC__)
__device__ inline u32 ldsLoad32(const u8 *base, u32 at)
{   // four bytes at any offset of a word-aligned LDS array: two aligned reads and a byte funnel shift
    const u32 *w = reinterpret_cast<const u32 *>(base) + (at >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], at & 3);
}

// histogram: counts[0..286) literal / length symbols, counts[286..316) distance symbols (no output); else: the member of block b in
// staging + b * DEFLATE_SLOT + DEFLATE_SLOT_PAD and its size in sizes[b]
__global__ void __launch_bounds__(64) k_deflate_blocks(const u8 *data, u64 nBytes, u64 firstBlock, u64 blockStride, u64 nBlocks, const DeflateTables *tables, const CrcConstants *crcConstants,
                                                       int histogram, unsigned long long *counts, u8 *staging, u32 *sizes)
{
    __shared__ __attribute__((aligned(16))) u8 in[DEFLATE_BLOCK_INPUT + 22];
    __shared__ u16 hashTable[1u << DEFLATE_HASH_BITS];
    // coding: the two code tables, the bit ring, the CRC table; counting: the histogram (in the same words)
    __shared__ u32 small[DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS + DEFLATE_RING_WORDS + 4 * 256];
    u32 *const litLen = small, *const distCode = small + DEFLATE_LITLEN_SYMBOLS, *const ring = distCode + DEFLATE_DIST_SYMBOLS, *const hist = small;
    u32 (*const crcTable)[256] = reinterpret_cast<u32 (*)[256]>(ring + DEFLATE_RING_WORDS);
    const u32 lane = threadIdx.x;
    const u64 block = firstBlock + u64(blockIdx.x) * blockStride;
    if (block >= nBlocks) return;
    const u64 from = block * DEFLATE_BLOCK_INPUT;
    const u32 n = u32(nBytes - from < DEFLATE_BLOCK_INPUT ? nBytes - from : DEFLATE_BLOCK_INPUT);
    const u8 *src = data + from;
    // stage the input (the blocks start on even addresses; every other one on a multiple of four)
    if (0 == (reinterpret_cast<u64>(src) & 3))
    {
        for (u32 i = 4 * lane; i + 4 <= n; i += 256) *reinterpret_cast<u32 *>(in + i) = *reinterpret_cast<const u32 *>(src + i);
        if (lane < (n & 3)) in[(n & ~3u) + lane] = src[(n & ~3u) + lane];
    }
    else for (u32 i = lane; i < n; i += 64) in[i] = src[i];
    for (u32 i = lane; i < (1u << DEFLATE_HASH_BITS); i += 64) hashTable[i] = 0xffff;
    if (histogram) { for (u32 i = lane; i < DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS; i += 64) hist[i] = 0; }
    else
    {
        for (u32 i = lane; i < DEFLATE_LITLEN_SYMBOLS; i += 64) litLen[i] = tables->litLen[i];
        if (lane < DEFLATE_DIST_SYMBOLS) distCode[lane] = tables->dist[lane];
        for (u32 i = lane; i < DEFLATE_RING_WORDS; i += 64) ring[i] = 0;
        for (u32 k = 0; k < 4; ++k) for (u32 i = lane; i < 256; i += 64) crcTable[k][i] = crcConstants->table[k][i];
    }
    __syncthreads();
    if (lane < 22) in[n + lane] = 0;               // the look-ahead reads past the end see zeros (matches are cut at n anyway)
    u32 *out = histogram ? nullptr : reinterpret_cast<u32 *>(staging + block * DEFLATE_SLOT + DEFLATE_SLOT_PAD + 18);       // word aligned: the slot is, 2 + 18 = 20
    u32 bitAt = 0, flushed = 0;                    // bits written so far / whole words already in global memory
    if (!histogram)
    {   // the dynamic block header
        const u32 headerBits = tables->headerBits, words = (headerBits + 31) / 32;
        for (u32 i = lane; i < words; i += 64) ring[i] = tables->header[i];
        bitAt = headerBits;
    }
    __syncthreads();
    u32 covered = 0;                               // positions below this are inside a match already taken
    for (u32 base = 0; base < n; base += 64)
    {
        const u32 p = base + lane;
        u32 length = 0, distance = 0;
        const u32 word = p < n ? ldsLoad32(in, p) : 0;
        if (p + DEFLATE_MIN_MATCH <= n)
        {
            const u32 h = (word * 2654435761u) >> (32 - DEFLATE_HASH_BITS);
            const u32 candidate = hashTable[h];
            hashTable[h] = u16(p);
            if (candidate != 0xffff && candidate < p && p - candidate <= DEFLATE_WINDOW && ldsLoad32(in, candidate) == word)
            {
                const u32 limit = n - p < DEFLATE_MAX_MATCH ? n - p : DEFLATE_MAX_MATCH;
                // sixteen bytes a turn: a lone wavefront waits out every LDS round trip, so each one should decide as much as it can
                u32 l = 4;
                while (l < limit)
                {
                    const u32 x0 = ldsLoad32(in, p + l) ^ ldsLoad32(in, candidate + l), x1 = ldsLoad32(in, p + l + 4) ^ ldsLoad32(in, candidate + l + 4);
                    const u32 x2 = ldsLoad32(in, p + l + 8) ^ ldsLoad32(in, candidate + l + 8), x3 = ldsLoad32(in, p + l + 12) ^ ldsLoad32(in, candidate + l + 12);
                    if (x0 | x1 | x2 | x3)
                    {
                        l += x0 ? u32(__ffs(int(x0)) - 1) >> 3 : x1 ? 4 + (u32(__ffs(int(x1)) - 1) >> 3) : x2 ? 8 + (u32(__ffs(int(x2)) - 1) >> 3) : 12 + (u32(__ffs(int(x3)) - 1) >> 3);
                        break;
                    }
                    l += 16;
                }
                length = l < limit ? l : limit; distance = p - candidate;
            }
        }
        // the greedy parse of these 64 positions
        u64 starts = __ballot(length != 0);
        u64 skipped = covered > base ? (covered - base >= 64 ? ~u64(0) : ((u64(1) << (covered - base)) - 1)) : 0;
        u64 taken = 0;
        while (true)
        {
            const u64 open = starts & ~skipped;
            if (!open) break;
            const u32 first = u32(__ffsll((unsigned long long)open)) - 1;
            const u32 l = u32(__shfl(int(length), int(first), 64));
            taken |= u64(1) << first;
            covered = base + first + l;
            const u32 end = first + l;                                 // lanes first + 1 .. end - 1 are inside the match
            const u64 below = end >= 64 ? ~u64(0) : ((u64(1) << end) - 1);
            skipped |= below & ~((u64(2) << first) - 1);
            starts &= ~((u64(2) << first) - 1);
        }
        const bool isMatch = (taken >> lane) & 1, isSkipped = (skipped >> lane) & 1, isLiteral = p < n && !isMatch && !isSkipped;
        if (histogram)
        {
            if (isLiteral) atomicAdd(&hist[word & 0xff], 1u);
            else if (isMatch)
            {
                u32 xb, xv;
                atomicAdd(&hist[257 + deflateLengthCode(length, xb, xv)], 1u);
                atomicAdd(&hist[DEFLATE_LITLEN_SYMBOLS + deflateDistanceCode(distance, xb, xv)], 1u);
            }
            continue;
        }
        u32 nBits = 0; u64 bits = 0;
        if (isLiteral) bits = deflateLiteralBits(litLen, word & 0xff, nBits);
        else if (isMatch) bits = deflateMatchBits(litLen, distCode, length, distance, nBits);
        u32 incl = nBits;
        for (u32 o = 1; o < 64; o <<= 1) { const u32 v = u32(__shfl_up(int(incl), o, 64)); if (lane >= o) incl += v; }
        const u32 total = u32(__shfl(int(incl), 63, 64));
        if (nBits)
        {
            const u32 at = bitAt + incl - nBits, w = at >> 5, s = at & 31;
            const u64 low = bits << s;                                  // 48 + 31 bits: up to three words
            atomicOr(&ring[w & (DEFLATE_RING_WORDS - 1)], u32(low));
            if (s + nBits > 32) atomicOr(&ring[(w + 1) & (DEFLATE_RING_WORDS - 1)], u32(low >> 32));
            if (s + nBits > 64) atomicOr(&ring[(w + 2) & (DEFLATE_RING_WORDS - 1)], u32(bits >> (64 - s)));
        }
        bitAt += total;
        __syncthreads();
        // whole words leave for global memory (as long as the block still fits its slot: a block that grows is stored instead)
        const u32 whole = bitAt >> 5;
        for (u32 w = flushed + lane; w < whole; w += 64)
        {
            if (4 * w + 4 <= DEFLATE_BLOCK_INPUT + 8) out[w] = ring[w & (DEFLATE_RING_WORDS - 1)];
            ring[w & (DEFLATE_RING_WORDS - 1)] = 0;
        }
        flushed = whole;
        __syncthreads();
    }
    if (histogram)
    {
        __syncthreads();
        for (u32 i = lane; i < DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS; i += 64) if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
        if (0 == lane) atomicAdd(&counts[DEFLATE_END_OF_BLOCK], 1ull);
        return;
    }
    // end of block, padding to a whole byte
    if (0 == lane)
    {
        const u32 c = litLen[DEFLATE_END_OF_BLOCK]; const u32 nBits = c >> 16, s = bitAt & 31, w = bitAt >> 5;
        const u64 low = u64(c & 0xffffu) << s;
        atomicOr(&ring[w & (DEFLATE_RING_WORDS - 1)], u32(low));
        if (s + nBits > 32) atomicOr(&ring[(w + 1) & (DEFLATE_RING_WORDS - 1)], u32(low >> 32));
    }
    bitAt += litLen[DEFLATE_END_OF_BLOCK] >> 16;
    __syncthreads();
    const u32 deflated = (bitAt + 7) / 8;          // bytes of deflate data
    const u32 lastWords = (deflated + 3) / 4;
    for (u32 w = flushed + lane; w < lastWords; w += 64) if (4 * w + 4 <= DEFLATE_BLOCK_INPUT + 8) out[w] = ring[w & (DEFLATE_RING_WORDS - 1)];
    // CRC-32 of the input: every lane a piece from a zero register, the pieces folded together (bgzf_kernels.h)
    const u32 piece = ((n + 63) / 64 + 3) & ~3u;
    const u32 begin = lane * piece < n ? lane * piece : n, end = begin + piece < n ? begin + piece : n;
    u32 crc = 0;
    u32 i = begin;
    for (; i + 4 <= end; i += 4)
    {
        const u32 w = crc ^ *reinterpret_cast<const u32 *>(in + i);
        crc = crcTable[3][w & 0xff] ^ crcTable[2][(w >> 8) & 0xff] ^ crcTable[1][(w >> 16) & 0xff] ^ crcTable[0][w >> 24];
    }
    for (; i < end; ++i) crc = crcTable[0][(crc ^ in[i]) & 0xff] ^ (crc >> 8);
    u32 lengthHere = end - begin;
    for (u32 step = 1; step < 64; step <<= 1)
    {
        const u32 otherCrc = u32(__shfl_down(int(crc), step, 64)), otherLength = u32(__shfl_down(int(lengthHere), step, 64));
        if (0 == (lane & (2 * step - 1)) && lane + step < 64 && otherLength)
        {
            crc = crcMultiply(crc, crcShiftOperator(crcConstants->squares, otherLength)) ^ otherCrc;
            lengthHere += otherLength;
        }
    }
    u8 *member = staging + block * DEFLATE_SLOT + DEFLATE_SLOT_PAD;
    const bool stored = deflated >= n;             // deflate did not shrink the block (or outgrew its slot): one stored deflate block instead
    __syncthreads();
    if (stored)
    {
        u8 *body = member + 18;
        if (0 == lane) { body[0] = 1; body[1] = u8(n); body[2] = u8(n >> 8); body[3] = u8(~n); body[4] = u8(~n >> 8); }
        for (u32 k = lane; k < n; k += 64) body[5 + k] = in[k];
    }
    const u32 bodyBytes = stored ? n + 5 : deflated;
    if (0 == lane)
    {
        const u32 value = ~(crcMultiply(0xffffffffu, crcShiftOperator(crcConstants->squares, n)) ^ crc);
        const u32 totalBytes = 18 + bodyBytes + 8, bsize = totalBytes - 1;
        const u8 header[18] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 0xff, 6, 0, 'B', 'C', 2, 0, u8(bsize), u8(bsize >> 8) };
        for (u32 k = 0; k < 18; ++k) member[k] = header[k];
        u8 *footer = member + 18 + bodyBytes;
        for (u32 k = 0; k < 4; ++k) { footer[k] = u8(value >> (8 * k)); footer[4 + k] = u8(n >> (8 * k)); }
        sizes[block] = totalBytes;
    }
}

// the members closed up: block b's bytes go to out + offsets[b]
__global__ void __launch_bounds__(256) k_deflate_gather(const u8 *staging, const u32 *sizes, const u64 *offsets, u64 nBlocks, u8 *out)
{
    const u64 b = blockIdx.x;
    if (b >= nBlocks) return;
    const u8 *src = staging + b * DEFLATE_SLOT + DEFLATE_SLOT_PAD;
    u8 *dst = out + offsets[b];
    const u32 n = sizes[b];
    // whole words on the destination's four-byte boundaries, single bytes before and after
    const u32 head = u32((4 - (reinterpret_cast<u64>(dst) & 3)) & 3) < n ? u32((4 - (reinterpret_cast<u64>(dst) & 3)) & 3) : n;
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    const u32 words = (n - head) / 4;
    for (u32 q = threadIdx.x; q < words; q += 256)
    {
        const u8 *s = src + head + 4 * q;
        *reinterpret_cast<u32 *>(dst + head + 4 * q) = u32(s[0]) | (u32(s[1]) << 8) | (u32(s[2]) << 16) | (u32(s[3]) << 24);
    }
    const u32 tail = n - head - 4 * words;
    if (threadIdx.x < tail) dst[head + 4 * words + threadIdx.x] = src[head + 4 * words + threadIdx.x];
}
#endif

} // namespace isaac
