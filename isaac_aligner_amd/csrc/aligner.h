// Thread-serial extend arithmetic: read decoding and trimming, CIGAR scan, ungapped / gapped / single-indel aligners and
// the per-cluster fragment builder.  One GPU thread runs these for one cluster; the 16-lane cooperative banded
// Smith-Waterman of bsw_kernel.h is the wavefront form of bswAlignSerial below.
//
// Behaviour follows (paths relative to /root/reference/src/c++):
//   lib/alignment/Read.cpp:32-73, Quality.cpp:72-120, fragmentBuilder/AlignerBase.cpp:50-227, UngappedAligner.cpp:39-92,
//   GappedAligner.cpp:51-82,167-249, BandedSmithWaterman.cpp:84-462, SimpleIndelAligner.cpp:50-518, FragmentBuilder.cpp:82-343
#pragma once
#include <cstring>
#include "types.h"
#include "sort.h"

namespace isaac
{

// ------------------------------------------------------------------------------------------------------------------
// reads straight from the BCL bytes (no decoded copy is kept): Read::decodeBcl semantics
struct ReadView
{
    const u8 *bcl;          // first cycle of this read
    u32 length;
    u32 endCyclesMasked;    // Read::endCyclesMasked_ (quality trimming), 0 when not trimmed
    u32 firstCycle;         // flowcell::ReadMetadata::getFirstCycle (1-based)
};
ISAAC_HD char strandBase(const ReadView &r, bool reverse, u32 i)
{
    const u8 b = reverse ? r.bcl[r.length - 1 - i] : r.bcl[i];
    if (!(b & 0xfc)) return 'n';                         // oligo::isBclN: any quality-0 byte is an N
    const u32 code = reverse ? (~u32(b)) & 3 : u32(b) & 3;
    return char(0x54474341u >> (8 * code));              // "ACGT"
}
ISAAC_HD u32 strandQuality(const ReadView &r, bool reverse, u32 i)
{
    const u8 b = reverse ? r.bcl[r.length - 1 - i] : r.bcl[i];
    return (b & 0xfc) ? u32(b >> 2) : 2u;
}
// include/alignment/Alignment.hh:44-47
ISAAC_HD bool isMatch(char readBase, char referenceBase) { return readBase == 'n' || (readBase == referenceBase && referenceBase != 'N'); }

// lib/alignment/Quality.cpp:72-105 (BWA-style 3' trimming); returns Read::endCyclesMasked_
ISAAC_HD u32 trimLowQualityEnd(const u8 *bcl, u32 length, u32 baseQualityCutoff)
{
    const u32 MASK_READ_LENGTH_MIN = 35;
    if (!baseQualityCutoff || length < MASK_READ_LENGTH_MIN) return 0;
    i32 qscoreSum = 0, peakSum = 0; bool trimPosSet = false; u32 trimPos = 0;
    for (u32 it = 0; length - MASK_READ_LENGTH_MIN != it; ++it)
    {
        const u8 b = bcl[length - 1 - it];
        const i32 q = (b & 0xfc) ? i32(b >> 2) : 2;
        qscoreSum += i32(baseQualityCutoff) - q;
        if (qscoreSum < 0) break;
        if (qscoreSum > peakSum) { peakSum = qscoreSum; trimPos = it; trimPosSet = true; }
    }
    return trimPosSet ? trimPos + 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// per-cluster cigar pool (alignment::Cigar buffer of FragmentBuilder / TemplateBuilder, fixed capacity here)
struct CigarPool
{
    u32 *words; u32 used; u32 capacity; u32 overflow;
    ISAAC_HD void push(u32 w) { if (used < capacity) words[used++] = w; else overflow = 1; }
    ISAAC_HD void addOperation(u32 len, u32 op) { push(cigarOp(len, op)); }
};

ISAAC_HD i64 candBeginClipped(const Cand &c, const u32 *pool)
{ return (c.cigarLength && OP_SOFT_CLIP == cigarCode(pool[c.cigarOffset])) ? i64(cigarLen(pool[c.cigarOffset])) : 0; }
ISAAC_HD i64 candEndClipped(const Cand &c, const u32 *pool)
{ const u32 w = pool[c.cigarOffset + (c.cigarLength ? c.cigarLength - 1 : 0)]; return (c.cigarLength && OP_SOFT_CLIP == cigarCode(w)) ? i64(cigarLen(w)) : 0; }
ISAAC_HD i64 candUnclippedPosition(const Cand &c, const u32 *pool) { return c.position - candBeginClipped(c, pool); }
ISAAC_HD u32 candMappedLength(const Cand &c, const u32 *pool)
{ u32 r = 0; for (u32 i = 0; i < c.cigarLength; ++i) if (OP_ALIGN == cigarCode(pool[c.cigarOffset + i])) r += cigarLen(pool[c.cigarOffset + i]); return r; }
// FragmentMetadata::resetAlignment + resetClipping (FragmentMetadata.hh:352-375)
ISAAC_HD void candResetAlignment(Cand &c, const CigarPool &pool)
{
    c.position = candUnclippedPosition(c, pool.words);
    c.cigarOffset = pool.used; c.cigarLength = 0; c.observedLength = 0; c.mismatchCount = 0; c.matchesInARow = 0; c.gapCount = 0; c.editDistance = 0;
    c.logProbability = 0.0; c.alignmentScore = 0xffffffffu; c.smithWatermanScore = 0;
}

// Sequential readers: the strand sequence of a read and the reference, 8 bytes per load instead of one.  A thread that
// walks 150 bases is bound by the latency of its loads, not by their size.
struct ReadStream
{
    const u8 *bcl; u32 length; bool reverse; u32 pos; u64 buf; u32 avail;
    ISAAC_HD void init(const ReadView &r, bool rev, u32 start) { bcl = r.bcl; length = r.length; reverse = rev; pos = start; avail = 0; buf = 0; }
    ISAAC_HD void refill()
    {
        const u32 left = pos < length ? length - pos : 0;
        if (left >= 8) { memcpy(&buf, reverse ? bcl + (length - pos - 8) : bcl + pos, 8); avail = 8; }
        else
        {
            buf = 0;
            for (u32 k = 0; k < left; ++k) { const u64 b = reverse ? bcl[length - 1 - pos - k] : bcl[pos + k]; buf |= reverse ? b << (8 * (7 - k)) : b << (8 * k); }
            avail = left ? left : 8;   // past the end (never for a valid CIGAR): zero bytes, i.e. 'n' with quality 2
        }
    }
    // the BCL byte of the next strand position; reverse strand: the caller complements (strandBaseOf)
    ISAAC_HD u8 next() { if (!avail) refill(); const u8 b = reverse ? u8(buf >> 56) : u8(buf); buf = reverse ? buf << 8 : buf >> 8; --avail; ++pos; return b; }
    ISAAC_HD void skip(u32 n) { if (n < avail) { buf = reverse ? buf << (8 * n) : buf >> (8 * n); avail -= n; } else avail = 0; pos += n; }
};
ISAAC_HD char strandBaseOf(u8 b, bool reverse)
{
    if (!(b & 0xfc)) return 'n';
    const u32 code = reverse ? (~u32(b)) & 3 : u32(b) & 3;
    return char(0x54474341u >> (8 * code));
}
ISAAC_HD u32 qualityOf(u8 b) { return (b & 0xfc) ? u32(b >> 2) : 2u; }
struct RefStream
{
    const char *p, *end; u64 buf; u32 avail;
    ISAAC_HD void init(const char *at, const char *basesEnd) { p = at; end = basesEnd; avail = 0; buf = 0; }
    ISAAC_HD char next()
    {
        if (!avail)
        {
            if (p + 8 <= end) { memcpy(&buf, p, 8); avail = 8; }
            else { buf = 0; u32 k = 0; for (; p + k < end; ++k) buf |= u64(u8(p[k])) << (8 * k); avail = k ? k : 8; }
        }
        const char c = char(u8(buf)); buf >>= 8; --avail; ++p;
        return c;
    }
    ISAAC_HD void skip(u32 n) { if (n < avail) { buf >>= 8 * n; avail -= n; } else avail = 0; p += n; }
};

// ---- eight bases of an ALIGN stretch at a time.  What depends on the order of the bases (the running fp64 sum, the length of
// the current run of matches) stays a loop over the eight; everything else -- strand bases, N handling, the comparison with the
// reference, the counters -- is done on all eight bytes of a 64-bit word at once.
static const u64 BYTES_01 = 0x0101010101010101ull;
// 0x80 in every byte of x that is zero, 0 elsewhere (exact: no carries between bytes)
ISAAC_HD u64 zeroBytes(u64 x) { const u64 m = 0x7f * BYTES_01; return ~(((x & m) + m) | x | m); }
// 'A' 'C' 'G' 'T' for the codes 0..3 in the bytes of `codes`
ISAAC_HD u64 asciiOfCodes(u64 codes)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // v_perm_b32 as a four-entry byte table: selector bytes 0..3 pick the bytes of the second operand
    const u32 lo = __builtin_amdgcn_perm(0u, 0x54474341u, u32(codes)), hi = __builtin_amdgcn_perm(0u, 0x54474341u, u32(codes >> 32));
    return u64(lo) | (u64(hi) << 32);
#else
    u64 r = 0;
    for (u32 k = 0; k < 8; ++k) r |= u64((0x54474341u >> (8 * u32((codes >> (8 * k)) & 3))) & 0xffu) << (8 * k);
    return r;
#endif
}
struct AlignBlock { u64 quality; u64 matchFlags; u64 differFlags; };   // byte k: strand position k of the block; flags are 0x80 or 0
// readBytes: the BCL bytes of eight consecutive strand positions (lowest position in the lowest byte); referenceBytes: the eight
// reference bytes under them.  isMatch (Alignment.hh:44-47), strandBaseOf and qualityOf for all eight.
ISAAC_HD AlignBlock alignBlock(u64 readBytes, bool reverse, u64 referenceBytes)
{
    AlignBlock b;
    const u64 nFlags = zeroBytes(readBytes & (0xfc * BYTES_01));                       // no quality bits: the base is an N
    u64 codes = readBytes & (0x03 * BYTES_01);
    if (reverse) codes ^= 0x03 * BYTES_01;
    const u64 nBytes = (nFlags >> 7) * 0xff;
    const u64 strand = (asciiOfCodes(codes) & ~nBytes) | ((0x6e * BYTES_01) & nBytes);   // 'n' for N
    const u64 equalFlags = zeroBytes(strand ^ referenceBytes);
    const u64 referenceNFlags = zeroBytes(referenceBytes ^ (0x4e * BYTES_01));
    b.matchFlags = nFlags | (equalFlags & ~referenceNFlags);
    b.differFlags = ~equalFlags & (0x80 * BYTES_01);
    b.quality = ((readBytes >> 2) & (0x3f * BYTES_01)) | (nFlags >> 6);                  // N: quality 2
    return b;
}
ISAAC_HD u64 loadBytes8(const void *p) { u64 v; memcpy(&v, p, 8); return v; }
// The same for four bases in 32 bits: what the scans use (a 64-bit shift, add or compare is two or more instructions on the device, and nothing
// here crosses from one byte to the next).  Flags are 0x80 or 0 per byte.
static const u32 QUAD_01 = 0x01010101u;
ISAAC_HD u32 zeroBytes32(u32 x) { const u32 m = 0x7f * QUAD_01; return ~(((x & m) + m) | x | m); }
ISAAC_HD u32 asciiOfCodes32(u32 codes)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, 0x54474341u, codes);
#else
    u32 r = 0;
    for (u32 k = 0; k < 4; ++k) r |= ((0x54474341u >> (8 * ((codes >> (8 * k)) & 3))) & 0xffu) << (8 * k);
    return r;
#endif
}
struct AlignQuad { u32 termAt; u32 matchFlags; u32 differFlags; };   // byte k: strand position k of the quad; termAt: quality, + 64 for a mismatch
// readBytes: the BCL bytes of four consecutive strand positions (lowest position in the lowest byte); complement: 0x03030303 for a reverse read, else 0
ISAAC_HD AlignQuad alignQuad(u32 readBytes, u32 complement, u32 referenceBytes)
{
    AlignQuad b;
    const u32 nFlags = zeroBytes32(readBytes & (0xfc * QUAD_01));                       // no quality bits: the base is an N
    // the strand's bytes from an eight-entry table: codes 0..3 are A C G T, 4..7 (an N: its flag moved to bit 2) are 'n'
    const u32 select = ((readBytes & (0x03 * QUAD_01)) ^ complement) | (nFlags >> 5);
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 strand = __builtin_amdgcn_perm(0x6e6e6e6eu, 0x54474341u, select);
#else
    u32 strand = 0;
    for (u32 k = 0; k < 4; ++k) { const u32 c = (select >> (8 * k)) & 7; strand |= (c < 4 ? (0x54474341u >> (8 * c)) & 0xffu : 0x6eu) << (8 * k); }
#endif
    const u32 equalFlags = zeroBytes32(strand ^ referenceBytes);
    // isMatch: an N of the read matches anything; otherwise equal bytes -- which are A C G or T, the strand has no upper-case N, so "and the
    // reference is not N" needs no test of its own
    b.matchFlags = nFlags | equalFlags;
    b.differFlags = ~equalFlags & (0x80 * QUAD_01);
    b.termAt = ((readBytes >> 2) & (0x3f * QUAD_01)) | (nFlags >> 6) | ((~b.matchFlags & (0x80 * QUAD_01)) >> 1);      // N: quality 2
    return b;
}
// the eight bytes of a block in strand order, as two quads: the bytes as loaded (forward), or last byte first (reverse)
ISAAC_HD void strandQuads(u64 loaded, bool reverse, u32 &first, u32 &second)
{
    const u32 lo = u32(loaded), hi = u32(loaded >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    first = __builtin_amdgcn_perm(hi, lo, reverse ? 0x04050607u : 0x03020100u);
    second = __builtin_amdgcn_perm(hi, lo, reverse ? 0x00010203u : 0x07060504u);
#else
    first = reverse ? __builtin_bswap32(hi) : lo; second = reverse ? __builtin_bswap32(lo) : hi;
#endif
}
ISAAC_HD u32 flagCount32(u32 flags)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return u32(__popc(flags));
#else
    return u32(__builtin_popcount(flags));
#endif
}
ISAAC_HD u32 flagCount(u64 flags)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return u32(__popcll(flags));
#else
    return u32(__builtin_popcountll(flags));
#endif
}
ISAAC_HD u64 reverseBytes(u64 x) { return __builtin_bswap64(x); }

// AlignerBase::updateFragmentCigar (AlignerBase.cpp:121-227).  logProbability is a running fp64 sum in base order: the
// order of the additions is part of the result, so this loop is deliberately serial.
ISAAC_HD u32 updateFragmentCigar(const DevParams &P, const DevReference &R, const ReadView &read, Cand &f, i64 strandPosition, const CigarPool &pool, u32 cigarOffset)
{
    const bool reverse = f.reverse;
    const char *reference = R.bases + R.contigOffset[f.contigId];
    f.cigarOffset = cigarOffset;
    f.cigarLength = u16(pool.used - cigarOffset);
    ReadStream rs; rs.init(read, reverse, 0);
    RefStream fs; fs.init(reference + strandPosition, R.bases + R.totalBases);
    u32 matchCount = 0;
    i64 referenceAdvance = 0;
    double lp = f.logProbability;
    u32 mismatchCount = f.mismatchCount, best = f.matchesInARow, editDistance = f.editDistance, gapCount = f.gapCount, sws = f.smithWatermanScore;
    for (u32 i = 0; f.cigarLength > i; ++i)
    {
        const u32 w = pool.words[cigarOffset + i];
        const u32 length = cigarLen(w), op = cigarCode(w);
        if (OP_ALIGN == op)
        {
            u32 matchesInARow = 0;
            u32 j = 0;
            // whole blocks of eight while the read and the reference both have eight bytes left.  Four blocks' loads are in flight ahead of the
            // arithmetic (a thread that asks for eight bytes, waits, and asks again spends the scan waiting: the kernels that do nothing but this
            // were issuing instructions a sixth of the time)
            u32 nBlocks = (length - j) / 8;
            nBlocks = imin(nBlocks, rs.pos <= rs.length ? (rs.length - rs.pos) / 8 : 0u);
            nBlocks = fs.p <= fs.end ? u32(imin(u64(nBlocks), u64(fs.end - fs.p) / 8)) : 0u;
            if (nBlocks)
            {
                const u8 *readAt = reverse ? rs.bcl + (rs.length - rs.pos - 8) : rs.bcl + rs.pos;       // block b: 8 b bytes further down (reverse) or up
                const char *referenceAt = fs.p;
                // (past the last block: the last block again, so that the load needs no branch; its bytes are not used)
                const auto load = [&](u32 b, u64 &readBytes, u64 &referenceBytes)
                {
                    const size_t at = 8 * size_t(b < nBlocks ? b : nBlocks - 1);
                    readBytes = loadBytes8(reverse ? readAt - at : readAt + at);
                    referenceBytes = loadBytes8(referenceAt + at);
                };
                const u32 complement = reverse ? 0x03 * QUAD_01 : 0u;
                // logMismatch behind logMatch in one table (the copies the kernels stage in LDS): a base's term is one read at quality + 64 * mismatch
                const bool joined = R.logMismatch == R.logMatch + 64 * R.logStride;
                const auto quad = [&](u32 readBytes, u32 referenceBytes)
                {
                    const AlignQuad q = alignQuad(readBytes, complement, referenceBytes);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (u32 k = 0; k < 4; ++k)
                    {
                        const u32 at = (q.termAt >> (8 * k)) & 0xffu;
                        lp += joined ? R.logMatch[at * R.logStride] : (at & 64 ? R.logMismatch[(at & 63) * R.logStride] : R.logMatch[at * R.logStride]);
                        matchesInARow = (matchesInARow + 1) & (0u - ((q.matchFlags >> (8 * k + 7)) & 1u));       // a mismatch: back to 0
                        best = imax(best, matchesInARow);
                    }
                    const u32 matches = flagCount32(q.matchFlags);
                    matchCount += matches; mismatchCount += 4 - matches;
                    editDistance += flagCount32(q.differFlags);
                };
                const auto block = [&](u64 readBytes, u64 referenceBytes)
                {
                    u32 first, second;
                    strandQuads(readBytes, reverse, first, second);
                    quad(first, u32(referenceBytes)); quad(second, u32(referenceBytes >> 32));
                };
                const u32 mismatchesBefore = mismatchCount;
#if !defined(ISAAC_SCAN_AHEAD)
#define ISAAC_SCAN_AHEAD 32
#endif
#if ISAAC_SCAN_AHEAD == 4
                u64 r0 = 0, r1 = 0, r2 = 0, r3 = 0, f0 = 0, f1 = 0, f2 = 0, f3 = 0;
                load(0, r0, f0); load(1, r1, f1); load(2, r2, f2); load(3, r3, f3);
                for (u32 b = 0; b < nBlocks; b += 4)
                {
                    block(r0, f0); load(b + 4, r0, f0);
                    if (b + 1 < nBlocks) { block(r1, f1); load(b + 5, r1, f1); }
                    if (b + 2 < nBlocks) { block(r2, f2); load(b + 6, r2, f2); }
                    if (b + 3 < nBlocks) { block(r3, f3); load(b + 7, r3, f3); }
                }
#elif ISAAC_SCAN_AHEAD == 2
                u64 r0 = 0, r1 = 0, f0 = 0, f1 = 0;
                load(0, r0, f0); load(1, r1, f1);
                for (u32 b = 0; b < nBlocks; b += 2)
                {
                    block(r0, f0); load(b + 2, r0, f0);
                    if (b + 1 < nBlocks) { block(r1, f1); load(b + 3, r1, f1); }
                }
#elif ISAAC_SCAN_AHEAD == 32
                // thirty-two bytes a load pair (measured against sixteen: see ISAAC_SCAN_AHEAD == 16)
                struct Bytes32 { u64 w0, w1, w2, w3; };
                const u32 nGroups = nBlocks / 4;
                const auto loadGroup = [&](u32 g, Bytes32 &readBytes, Bytes32 &referenceBytes)
                {
                    const size_t at = 32 * size_t(g < nGroups ? g : nGroups - 1);
                    memcpy(&readBytes, reverse ? readAt - at - 24 : readAt + at, 32);
                    memcpy(&referenceBytes, referenceAt + at, 32);
                };
                if (nGroups)
                {
                    Bytes32 ra, fa, rb, fb;
                    loadGroup(0, ra, fa);
                    for (u32 g = 0; g < nGroups; ++g)
                    {
                        loadGroup(g + 1, rb, fb);
                        block(reverse ? ra.w3 : ra.w0, fa.w0); block(reverse ? ra.w2 : ra.w1, fa.w1); block(reverse ? ra.w1 : ra.w2, fa.w2); block(reverse ? ra.w0 : ra.w3, fa.w3);
                        ra = rb; fa = fb;
                    }
                }
                for (u32 b = 4 * nGroups; b < nBlocks; ++b) { u64 r0 = 0, f0 = 0; load(b, r0, f0); block(r0, f0); }
#elif ISAAC_SCAN_AHEAD == 16
                // sixteen bytes a load, the next pair of blocks on its way while the current pair is worked on; an odd last block by itself
                struct Bytes16 { u64 lo, hi; };
                const u32 nPairs = nBlocks / 2;
                const auto loadPair = [&](u32 g, Bytes16 &readBytes, Bytes16 &referenceBytes)
                {
                    const size_t at = 16 * size_t(g < nPairs ? g : nPairs - 1);
                    memcpy(&readBytes, reverse ? readAt - at - 8 : readAt + at, 16);
                    memcpy(&referenceBytes, referenceAt + at, 16);
                };
                if (nPairs)
                {
                    Bytes16 ra, fa, rb, fb;
                    loadPair(0, ra, fa);
                    for (u32 g = 0; g < nPairs; ++g)
                    {
                        loadPair(g + 1, rb, fb);
                        // reverse: the pair's first block is the upper half of what was loaded
                        block(reverse ? ra.hi : ra.lo, fa.lo); block(reverse ? ra.lo : ra.hi, fa.hi);
                        ra = rb; fa = fb;
                    }
                }
                if (nBlocks & 1) { u64 r0 = 0, f0 = 0; load(nBlocks - 1, r0, f0); block(r0, f0); }
#else
                u64 r0 = 0, f0 = 0, r1 = 0, f1 = 0;
                load(0, r0, f0);
                for (u32 b = 0; b < nBlocks; ++b) { load(b + 1, r1, f1); block(r0, f0); r0 = r1; f0 = f1; }
#endif
                sws += (mismatchCount - mismatchesBefore) * P.normalizedMismatchScore;
                rs.pos += 8 * nBlocks; rs.avail = 0; fs.p += 8 * size_t(nBlocks); fs.avail = 0;
                j += 8 * nBlocks;
            }
            for (; length > j; ++j)
            {
                const u8 b = rs.next();
                const char s = strandBaseOf(b, reverse);
                const u32 q = qualityOf(b);
                const char r = fs.next();
                if (isMatch(s, r)) { ++matchCount; ++matchesInARow; lp += R.logMatch[q * R.logStride]; }
                else
                {
                    best = imax(best, matchesInARow); matchesInARow = 0;
                    ++mismatchCount; lp += R.logMismatch[q * R.logStride]; sws += P.normalizedMismatchScore;
                }
                if (s != r) ++editDistance;
            }
            referenceAdvance += length;
            best = imax(best, matchesInARow);
        }
        else if (OP_INSERT == op)
        {
            rs.skip(length); editDistance += length; ++gapCount;
            sws += P.normalizedGapOpenScore + imin(P.normalizedMaxGapExtendScore, (length - 1) * P.normalizedGapExtendScore);
        }
        else if (OP_DELETE == op)
        {
            fs.skip(length); referenceAdvance += length; editDistance += length; ++gapCount;
            sws += P.normalizedGapOpenScore + imin(P.normalizedMaxGapExtendScore, (length - 1) * P.normalizedGapExtendScore);
        }
        else // OP_SOFT_CLIP
        {
            for (u32 j = 0; j < length; ++j) lp += R.logMatch[qualityOf(rs.next()) * R.logStride];
        }
    }
    f.logProbability = lp; f.mismatchCount = u16(mismatchCount); f.matchesInARow = u16(best); f.editDistance = u16(editDistance);
    f.gapCount = u16(gapCount); f.smithWatermanScore = sws;
    f.observedLength = u32(referenceAdvance);
    f.position = strandPosition;
    return matchCount;
}

// ------------------------------------------------------------------------------------------------------------------
// Sequencing adapters: matchSelector::SequencingAdapter::getMatchRange (lib/alignment/matchSelector/SequencingAdapter.cpp:58-141) and
// FragmentSequencingAdapterClipper (lib/alignment/matchSelector/FragmentSequencingAdapterClipper.cpp:41-282) on offsets into the strand sequence.
// A strand's adapter range travels as one word: begin | end << 16, 0 = no adapter on this strand (a found range has begin < end).
ISAAC_HD u32 adapterRangePack(i64 begin, i64 end) { return u32(begin) | (u32(end) << 16); }
// oligo::generateKmer (KmerGenerator.hpp:149-169) for the 5 bases from `at`: anything but ACGT has the value 4, which spills into the base before it
ISAAC_HD bool adapterKmer(const ReadView &read, bool reverse, i64 at, i64 end, u32 &kmer)
{
    u32 k = 0;
    for (u32 todo = ADAPTER_MATCH_BASES_MIN; todo; --todo, ++at)
    {
        if (at == end) return false;
        const char c = strandBase(read, reverse, u32(at));
        const u32 v = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
        k = ((k << 2) | v) & 0xffffu;                 // (the reference's k-mer is an unsigned short)
    }
    kmer = k & ((1u << (2 * ADAPTER_MATCH_BASES_MIN)) - 1);
    return true;
}
ISAAC_HD void adapterMatchRange(const DevAdapter &a, const ReadView &read, bool reverse, i64 sequenceBegin, i64 sequenceEnd, i64 mismatchBase, i64 &first, i64 &second)
{
    first = second = mismatchBase;
    u32 kmer;
    if (!adapterKmer(read, reverse, mismatchBase, sequenceEnd, kmer)) return;
    const i32 pos = a.kmerPositions[kmer];
    if (pos < 0) return;
    const bool unbounded = 0 == a.clipLength;
    const u32 mismatchBaseOffset = u32(mismatchBase - sequenceBegin);
    const u32 adapterBasesBeforeSequence = mismatchBaseOffset < u32(pos) ? u32(pos) - mismatchBaseOffset : 0;
    if (adapterBasesBeforeSequence && unbounded) return;
    const i64 testBase = mismatchBase - (i64(pos) - i64(adapterBasesBeforeSequence));
    const u32 testSequenceLength = u32(sequenceEnd - testBase);
    const u32 leftClippedAdapterLength = a.length - adapterBasesBeforeSequence;
    const u32 overlapLength = imin(testSequenceLength, leftClippedAdapterLength);
    if (overlapLength < leftClippedAdapterLength && unbounded && a.reverse) return;
    if (overlapLength < ADAPTER_MATCH_BASES_MIN) return;
    for (u32 i = 0; i < overlapLength; ++i) if (a.sequence[adapterBasesBeforeSequence + i] != strandBase(read, reverse, u32(testBase) + i)) return;
    if (a.reverse)
    {
        first = unbounded ? sequenceBegin : testBase - i64(imin(u32(testBase - sequenceBegin), a.clipLength - a.length));
        second = testBase + overlapLength;
    }
    else
    {
        first = testBase;
        second = unbounded ? sequenceEnd : testBase + i64(imin(overlapLength, a.clipLength));
    }
}
// checkInitStrand (:103-150) for a fragment of `read` on strand `reverse` at `position` of the contig: where the strand's adapter lies
ISAAC_HD u32 adapterStrandRange(const DevAdapters &A, const DevReference &R, const ReadView &read, bool reverse, u32 contigId, i64 position)
{
    const i64 referenceSize = i64(contigLength(R, contigId));
    i64 sequenceBegin = 0, sequenceEnd = read.length;
    const i64 referenceLeft = referenceSize - position;
    if (referenceLeft < sequenceEnd - sequenceBegin) sequenceEnd = sequenceBegin + referenceLeft;
    i64 newFragmentPos = position;
    if (0 > position) { sequenceBegin -= position; newFragmentPos = 0; }
    i64 rangeBegin = sequenceEnd, rangeEnd = sequenceBegin;
    const char *reference = R.bases + R.contigOffset[contigId];
    for (u32 k = 0; k < A.n; ++k)
    {
        const DevAdapter &a = A.a[k];
        if (0 == a.clipLength && reverse != (0 != a.reverse)) continue;           // isStrandCompatible
        if (rangeEnd >= sequenceEnd) continue;
        // findSequencingAdapter (:76-97) from where the adapters found so far end
        const i64 searchBegin = rangeEnd;
        i64 currentReference = newFragmentPos + (searchBegin - sequenceBegin);
        for (i64 currentBase = searchBegin; currentBase != sequenceEnd; ++currentBase, ++currentReference)
        {
            if (isMatch(strandBase(read, reverse, u32(currentBase)), reference[currentReference])) continue;
            i64 first, second;
            adapterMatchRange(a, read, reverse, searchBegin, sequenceEnd, currentBase, first, second);
            if (first != second) { rangeBegin = imin(first, rangeBegin); rangeEnd = imax(second, rangeEnd); break; }
        }
    }
    return sequenceBegin == rangeEnd ? 0u : adapterRangePack(rangeBegin, rangeEnd);
}
// countMatches / countMismatches (Alignment.hh:91-146) for decideWhichSideToClip; a reference index outside the contig compares as 'N' (the reference
// forms reference.begin() + contigPosition unchecked -- undefined for a fragment that hangs over the contig's start and is not the strand's first; the
// oracle defines it the same way)
ISAAC_HD u32 adapterCountMatches(const ReadView &read, bool reverse, i64 sequenceBegin, i64 sequenceEnd, const char *reference, i64 referenceSize, i64 referenceBegin, i64 referenceEnd, bool matches)
{
    u32 ret = 0;
    for (; sequenceEnd != sequenceBegin && referenceEnd != referenceBegin; ++sequenceBegin, ++referenceBegin)
    {
        const char ref = (referenceBegin < 0 || referenceBegin >= referenceSize) ? 'N' : reference[referenceBegin];
        ret += matches == isMatch(strandBase(read, reverse, u32(sequenceBegin)), ref);
    }
    return ret;
}
// FragmentSequencingAdapterClipper::clip with decideWhichSideToClip (:152-278): [begin, end) is the whole strand sequence on entry
ISAAC_HD void adapterClip(const DevReference &R, const ReadView &read, Cand &f, u32 range, i64 &begin, i64 &end)
{
    if (!range) return;
    const i64 rangeBegin = range & 0xffffu, rangeEnd = range >> 16, sequenceLength = read.length;
    const bool reverse = 0 != f.reverse;
    const u32 backwardsClipped = u32(rangeBegin), forwardsClipped = u32(sequenceLength - rangeEnd);
    bool clipBackwards = backwardsClipped < forwardsClipped, doClip = true;
    const char *reference = R.bases + R.contigOffset[f.contigId];
    const i64 referenceSize = i64(contigLength(R, f.contigId)), contigPosition = f.position;
    const i32 difference = i32(backwardsClipped - forwardsClipped);
    if (backwardsClipped && forwardsClipped && (difference < 0 ? -difference : difference) < 9)
    {
        if (contigPosition >= 0 && u64(referenceSize) >= u64(u32(contigPosition + sequenceLength)))
        {
            const i64 referenceBegin = contigPosition, referenceEnd = referenceBegin + sequenceLength;
            const u32 backwardsMatches = adapterCountMatches(read, reverse, 0, rangeBegin, reference, referenceSize, referenceBegin, referenceBegin + backwardsClipped, true);
            const u32 forwardsMatches = adapterCountMatches(read, reverse, rangeEnd, sequenceLength, reference, referenceSize, referenceEnd - forwardsClipped, referenceEnd, true);
            clipBackwards = backwardsMatches < forwardsMatches || (backwardsMatches == forwardsMatches && backwardsClipped < forwardsClipped);
        }
    }
    else if (!backwardsClipped || !forwardsClipped)
    {
        const i64 referenceBegin = contigPosition, referenceEnd = referenceBegin + sequenceLength;
        const u32 TOO_GOOD_READ_MISMATCH_PERCENT = 40;
        if (clipBackwards && !backwardsClipped)
        {
            const u32 basesClipped = u32(rangeEnd);
            doClip = adapterCountMatches(read, reverse, 0, rangeEnd, reference, referenceSize, referenceBegin, referenceBegin + basesClipped, false) * 100 / basesClipped > TOO_GOOD_READ_MISMATCH_PERCENT;
        }
        else if (!clipBackwards && !forwardsClipped)
        {
            const u32 basesClipped = u32(sequenceLength - rangeBegin);
            doClip = adapterCountMatches(read, reverse, rangeBegin, sequenceLength, reference, referenceSize, referenceEnd - basesClipped, referenceEnd, false) * 100 / basesClipped > TOO_GOOD_READ_MISMATCH_PERCENT;
        }
    }
    if (!doClip) return;
    if (clipBackwards) { candIncrementClipLeft(f, u32(rangeEnd)); begin = rangeEnd; }
    else { candIncrementClipRight(f, u32(sequenceLength - rangeBegin)); end = rangeBegin; }
}

// FragmentSequencingAdapterClipper::clip + AlignerBase::clipReadMasking + clipReference (UngappedAligner.cpp:59-62, AlignerBase.cpp:50-119) on [begin, end)
// indices of the strand sequence.  adapterRange: the strand's adapter (0: none; R is then not looked at).
// Returns false for the "position past the contig end" case (AlignerBase.cpp:74-81), which no seed or rescue candidate can
// produce (positions are always < contig length) and which is undefined behaviour in the reference.
ISAAC_HD bool clipSequence(const ReadView &read, Cand &f, i64 referenceSize, i64 &begin, i64 &end, const DevReference *R = 0, u32 adapterRange = 0)
{
    begin = 0; end = read.length;
    if (adapterRange) adapterClip(*R, read, f, adapterRange, begin, end);
    const i64 maskedBegin = f.reverse ? i64(read.endCyclesMasked) : 0;
    const i64 maskedEnd = f.reverse ? i64(read.length) : i64(read.length) - i64(read.endCyclesMasked);
    if (maskedBegin > begin) { candIncrementClipLeft(f, u32(maskedBegin - begin)); begin = maskedBegin; }
    if (maskedEnd < end) { candIncrementClipRight(f, u32(end - maskedEnd)); end = maskedEnd; }
    const i64 referenceLeft = referenceSize - f.position;
    if (referenceLeft < 0) return false;
    if (referenceLeft < end - begin) end = begin + referenceLeft;
    if (0 > f.position) { begin -= f.position; f.position = 0; }
    end = imax(end, begin);
    return true;
}

// UngappedAligner::alignUngapped (UngappedAligner.cpp:39-92)
ISAAC_HD u32 alignUngapped(const DevParams &P, const DevReference &R, const ReadView &read, Cand &f, CigarPool &pool, u32 adapterRange = 0)
{
    const u32 cigarOffset = pool.used;
    candResetAlignment(f, pool);
    f.lowClipped = 0; f.highClipped = 0;
    i64 begin, end;
    if (!clipSequence(read, f, i64(contigLength(R, f.contigId)), begin, end, &R, adapterRange)) { candSetUnaligned(f); return 0; }
    if (begin > i64(read.length)) { candSetUnaligned(f); return 0; }
    if (begin) pool.addOperation(u32(begin), OP_SOFT_CLIP);
    if (end - begin) pool.addOperation(u32(end - begin), OP_ALIGN);
    if (i64(read.length) - end) pool.addOperation(u32(i64(read.length) - end), OP_SOFT_CLIP);
    const u32 ret = updateFragmentCigar(P, R, read, f, f.position, pool, cigarOffset);
    if (!ret) candSetUnaligned(f);
    return ret;
}

// ------------------------------------------------------------------------------------------------------------------
// BandedSmithWaterman::align (BandedSmithWaterman.cpp:84-462), one thread, the 16 lanes of the SSE registers as arrays.
// The traceback flags are packed 2 bits each: 3 words per query row (TG, TE, TF) in `tflags` (3 * queryLength words).
static const u32 BSW_WIDEST_GAP_SIZE = 16, BSW_DISTANCE_CUTOFF = 7, BSW_MISMATCHES_CUTOFF = 5;
ISAAC_HD i16 w16(i32 v) { return i16(u16(v)); }

template <typename QueryF>
ISAAC_HD u32 bswAlignSerial(const DevParams &P, QueryF query, u32 querySize, const char *database, u32 *tflags, CigarPool &cigar)
{
    const i32 matchScore = P.gapMatch, mismatchScore = P.gapMismatch, gapOpenScore = -P.gapOpen, gapExtendScore = -P.gapExtend; // GappedAligner.cpp:41-42
    const i16 initialValue = i16(-32768 + gapOpenScore);
    const i16 open = i16(gapOpenScore), ext = i16(gapExtendScore);
    const u32 originalCigarSize = cigar.used;
    i16 E[16], F[16], G[16];
    for (u32 k = 0; k < 16; ++k) { E[k] = initialValue; F[k] = 0; G[k] = initialValue; }
    G[0] = 0;
    for (u32 i = 0; i < querySize; ++i)
    {
        i16 newF[16], newG[16];
        u32 TF = 0, TG = 0, TE = 0;
        for (u32 k = 1; k < 16; ++k)
        {
            const i16 g = G[k - 1], e = E[k - 1];
            u32 tf = (g < e) ? 1 : 0;
            const i16 v = w16(i32(imax(g, e)) - open);
            const i16 fe = w16(i32(F[k - 1]) - ext);
            if (v < fe) tf = 2;
            newF[k] = imax(v, fe);
            TF |= tf << (2 * k);
        }
        newF[0] = initialValue;
        u32 fE = 0, fF = 0; // one bit per lane
        for (u32 k = 0; k < 16; ++k)
        {
            if (G[k] < E[k]) fE |= 1u << k;
            const i16 m = imax(G[k], E[k]);
            if (m < F[k]) fF |= 1u << k;
            newG[k] = imax(m, F[k]);
        }
        // _mm_max_epi16 over (even, odd) byte pairs of the F-flags (0/2) and E-flags (0/1): BandedSmithWaterman.cpp:197
        for (u32 m = 0; m < 8; ++m)
        {
            const u32 e0 = (fE >> (2 * m)) & 1, e1 = (fE >> (2 * m + 1)) & 1, f0 = (fF >> (2 * m)) & 1, f1 = (fF >> (2 * m + 1)) & 1;
            const u32 a = (2 * f1) * 256 + 2 * f0, b = e1 * 256 + e0;
            const u32 r = a > b ? a : b;
            TG |= (r & 0xff) << (4 * m); TG |= (r >> 8) << (4 * m + 2);
        }
        const char q = query(i);
        for (u32 k = 0; k < 16; ++k)
        {
            const bool diff = q != database[i + 15 - k];
            const u16 w = diff ? u16(0xff00u | u8(mismatchScore)) : u16(u8(matchScore));
            newG[k] = w16(i32(newG[k]) + i32(i16(w)));
        }
        {
            i16 g = initialValue, e = initialValue, f = initialValue;
            for (u32 j = 0; j < 16; ++j)
            {
                const u32 k = 15 - j;
                i16 mx = g; u32 tMax = 0;
                if (e > g && e > f) { mx = e; tMax = 1; }
                else if (f > g) { mx = f; tMax = 2; }
                TE |= tMax << (2 * k);
                E[k] = mx;
                g = w16(i32(newG[k]) - open);
                e = w16(i32(mx) - gapExtendScore);
                f = w16(i32(newF[k]) - open);
            }
        }
        for (u32 k = 0; k < 16; ++k) { G[k] = newG[k]; F[k] = newF[k]; }
        tflags[3 * i] = TG; tflags[3 * i + 1] = TE; tflags[3 * i + 2] = TF;
    }
    i16 mx = w16(i32(u16(G[15])) - 1);
    i32 ii = i32(querySize) - 1, jj = ii; u32 maxType = 0;
    for (i32 k = 15; k >= 0; --k)
    {
        if (G[k] > mx) { mx = G[k]; jj = k; maxType = 0; }
        if (E[k] > mx) { mx = E[k]; jj = k; maxType = 1; }
        if (F[k] > mx) { mx = F[k]; jj = k; maxType = 2; }
    }
    u32 opLength = 0;
    if (jj > 0) cigar.addOperation(u32(jj), OP_DELETE);
    while (ii >= 0 && jj >= 0 && jj <= 15)
    {
        ++opLength;
        const u32 nextMaxType = (tflags[3 * ii + maxType] >> (2 * jj)) & 3;
        const u32 op = maxType == 0 ? OP_ALIGN : maxType == 1 ? OP_DELETE : OP_INSERT;
        if (nextMaxType != maxType) { cigar.addOperation(opLength, op); opLength = 0; }
        ii += (maxType == 1) ? 0 : -1;
        jj += (maxType == 1) ? 1 : (maxType == 2) ? -1 : 0;
        maxType = nextMaxType;
    }
    if (1 != maxType && opLength) { cigar.addOperation(opLength, maxType == 0 ? OP_ALIGN : OP_INSERT); opLength = 0; }
    if (15 > jj) { cigar.addOperation(opLength + 15 - u32(jj), OP_DELETE); opLength = 0; }
    u32 ret = 0;
    if (cigar.used > originalCigarSize && OP_DELETE == cigarCode(cigar.words[cigar.used - 1])) { ret = cigarLen(cigar.words[cigar.used - 1]); --cigar.used; }
    for (u32 lo = originalCigarSize, hi = cigar.used; lo + 1 < hi; ++lo) { --hi; const u32 t = cigar.words[lo]; cigar.words[lo] = cigar.words[hi]; cigar.words[hi] = t; }
    if (cigar.used > originalCigarSize && OP_DELETE == cigarCode(cigar.words[cigar.used - 1])) --cigar.used;
    return ret;
}

// GappedAligner::getFlanks (GappedAligner.cpp:51-82)
ISAAC_HD void getFlanks(i64 strandPosition, u32 readLength, u64 referenceSize, u32 &left, u32 &right)
{
    const u32 w = BSW_WIDEST_GAP_SIZE;
    if (strandPosition >= i64(w / 2))
    {
        if (strandPosition + i64(readLength) + i64(w - w / 2) < i64(referenceSize)) { left = w / 2; right = w - left - 1; }
        else { right = u32(referenceSize - readLength - u64(strandPosition)); left = w - right - 1; }
    }
    else { left = u32(strandPosition); right = w - left - 1; }
}

struct StrandQuery
{
    const ReadView *read; bool reverse; u32 offset;
    ISAAC_HD char operator()(u32 i) const { return strandBase(*read, reverse, offset + i); }
};

// GappedAligner::alignGapped (GappedAligner.cpp:167-249), --avoid-smith-waterman 0
ISAAC_HD u32 alignGapped(const DevParams &P, const DevReference &R, const ReadView &read, Cand &f, CigarPool &pool, u32 *tflags, u32 adapterRange = 0)
{
    const u32 cigarOffset = pool.used;
    candResetAlignment(f, pool);
    f.lowClipped = 0; f.highClipped = 0;
    const u64 referenceSize = contigLength(R, f.contigId);
    i64 begin, end;
    if (!clipSequence(read, f, i64(referenceSize), begin, end, &R, adapterRange)) return 0;
    if (begin) pool.addOperation(u32(begin), OP_SOFT_CLIP);
    const u32 sequenceLength = u32(end - begin);
    i64 strandPosition = f.position;
    if (i64(referenceSize) < i64(sequenceLength) + strandPosition + i64(BSW_WIDEST_GAP_SIZE)) return 0;
    if (!sequenceLength) return 0; // the reference would read cigar.back() of an empty alignment here; cannot happen for seeded candidates
    u32 left, right;
    getFlanks(strandPosition, sequenceLength, referenceSize, left, right);
    const char *database = R.bases + R.contigOffset[f.contigId] + strandPosition - left;
    StrandQuery q; q.read = &read; q.reverse = f.reverse; q.offset = u32(begin);
    strandPosition += bswAlignSerial(P, q, sequenceLength, database, tflags, pool);
    const u32 clipEndBases = u32(i64(read.length) - end);
    if (clipEndBases) pool.addOperation(clipEndBases, OP_SOFT_CLIP);
    strandPosition -= left;
    return updateFragmentCigar(P, R, read, f, strandPosition, pool, cigarOffset);
}

// ------------------------------------------------------------------------------------------------------------------
// SimpleIndelAligner (SimpleIndelAligner.cpp:50-438)
static const u32 GAP_FLANK_BASES = 32, GAP_FLANK_MISMATCHES_MAX = 8;

// a contig, or a copy of a stretch of it that is addressed with the contig's own offsets (k_indel_fragments keeps one in LDS)
struct RefView { const char *p; i64 base; ISAAC_HD char operator[](i64 i) const { return p[i - base]; } };

// countMismatches(seq + seqOffset, ref + refOffset .. refEnd, length) of Alignment.hh:115-157
// `lanes` > 1 (wave-per-cluster form, every lane calling with the same arguments): the positions are spread over the lanes and
// the counts summed across the wave, so every lane returns the total
ISAAC_HD u32 countMismatches(const ReadView &read, bool reverse, i64 seqOffset, const RefView &reference, i64 refOffset, i64 refSize, u32 length, u32 lanes = 1, u32 lane = 0)
{
    u32 ret = 0;
    for (u32 i = lane; i < length && refOffset + i64(i) < refSize; i += lanes) ret += !isMatch(strandBase(read, reverse, u32(seqOffset + i)), reference[refOffset + i]);
#if defined(__HIP_DEVICE_COMPILE__)
    if (lanes > 1) for (int o = 32; o > 0; o >>= 1) ret += __shfl_xor(ret, o, 64);
#endif
    return ret;
}

// `referenceOverride`: a copy of the contig around the two candidates, addressed like the contig itself (k_indel_fragments
// stages it in LDS); the CIGAR rescan of an accepted indel reads the real contig
ISAAC_HD void alignSimpleDeletion(const DevParams &P, const DevReference &R, const ReadView &read, CigarPool &pool, Cand &head, u32 headSeedOffset,
                                  Cand &tail, u32 tailSeedOffset, u32 tailSeedLength, u32 &simpleIndels, const RefView *referenceOverride = 0, const ReadView *scanRead = 0, u32 lanes = 1, u32 lane = 0)
{
    const u32 *cw = pool.words;
    if (i64(headSeedOffset) < candBeginClipped(head, cw)) return;
    if (candBeginClipped(tail, cw) + i64(candObservedLength(tail)) < i64(tailSeedOffset + tailSeedLength)) return;
    const u32 tailOffset = headSeedOffset;
    const bool reverse = head.reverse;
    RefView reference; if (referenceOverride) reference = *referenceOverride; else { reference.p = R.bases + R.contigOffset[head.contigId]; reference.base = 0; }
    const i64 refSize = i64(contigLength(R, head.contigId));
    const i64 headUnclipped = candUnclippedPosition(head, cw), tailUnclipped = candUnclippedPosition(tail, cw);
    i64 tailIt = tailOffset;                                          // index into the strand sequence
    u32 tailLength = u32(candBeginClipped(tail, cw) + i64(candObservedLength(tail)) - i64(tailOffset));
    const u32 tailMismatches = countMismatches(read, reverse, tailIt, reference, headUnclipped + tailOffset, refSize, tailLength, lanes, lane);
    if (!tailMismatches) return;
    const i64 deletionLengthL = tailUnclipped - headUnclipped;
    if (deletionLengthL < 0) return;                                  // boost::numeric_cast would throw; unreachable for ordered lists
    const u32 deletionLength = u32(deletionLengthL);
    u32 rightRealignedMismatches = countMismatches(read, reverse, tailIt, reference, tailUnclipped + tailOffset, refSize, tailLength, lanes, lane);
    u32 leftRealignedMismatches = 0;
    const u32 lf = imin(GAP_FLANK_BASES, tailOffset);
    u32 leftFlankMismatches = countMismatches(read, reverse, tailIt - lf, reference, headUnclipped + tailOffset - imin(32u, tailOffset), refSize, lf, lanes, lane);
    u32 rightFlankMismatches = countMismatches(read, reverse, tailIt, reference, tailUnclipped + tailOffset, refSize, imin(GAP_FLANK_BASES, tailLength), lanes, lane);
    i64 refIt = headUnclipped + tailOffset;
    u32 bestMismatches = tailMismatches, bestLeftFlankMismatches = leftFlankMismatches, bestRightFlankMismatches = rightFlankMismatches;
    u32 bestOffset = 0xffffffffu;
    for (u32 deletionOffset = tailOffset; bestMismatches && deletionOffset <= tailSeedOffset; ++deletionOffset, ++tailIt, ++refIt, --tailLength)
    {
        const u32 thisOffsetMismatches = leftRealignedMismatches + rightRealignedMismatches;
        if (bestMismatches > thisOffsetMismatches)
        {
            bestOffset = deletionOffset; bestMismatches = thisOffsetMismatches;
            bestLeftFlankMismatches = leftFlankMismatches; bestRightFlankMismatches = rightFlankMismatches;
        }
        const char tb = strandBase(read, reverse, u32(tailIt));
        const bool newLeftMismatch = !isMatch(tb, reference[refIt]);
        leftRealignedMismatches += newLeftMismatch; leftFlankMismatches += newLeftMismatch;
        if (deletionOffset >= GAP_FLANK_BASES)
            leftFlankMismatches -= !isMatch(strandBase(read, reverse, u32(tailIt - GAP_FLANK_BASES)), reference[refIt - GAP_FLANK_BASES]);
        const bool disappearingRightMismatch = !isMatch(tb, reference[refIt + deletionLength]);
        rightRealignedMismatches -= disappearingRightMismatch; rightFlankMismatches -= disappearingRightMismatch;
        if (tailLength > GAP_FLANK_BASES)
            rightFlankMismatches += !isMatch(strandBase(read, reverse, u32(tailIt + GAP_FLANK_BASES)), reference[refIt + deletionLength + GAP_FLANK_BASES]);
    }
    if (bestLeftFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && bestRightFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && 0xffffffffu != bestOffset)
    {
        const i64 clippingPositionOffset = candBeginClipped(head, cw);
        const u32 leftMapped = u32(i64(bestOffset) - clippingPositionOffset);
        const u32 headMismatches = countMismatches(read, reverse, clippingPositionOffset, reference, head.position, refSize, leftMapped, lanes, lane);
        const u32 newMismatches = headMismatches + bestMismatches;
        const u32 sws = P.normalizedMismatchScore * newMismatches + P.normalizedGapOpenScore +
            imin(P.normalizedMaxGapExtendScore, (deletionLength - 1) * P.normalizedGapExtendScore);
        if (head.smithWatermanScore > sws || (head.smithWatermanScore == sws && head.mismatchCount > newMismatches))
        {
            const u32 cigarOffset = pool.used;
            if (clippingPositionOffset) pool.addOperation(u32(clippingPositionOffset), OP_SOFT_CLIP);
            if (leftMapped) { pool.addOperation(leftMapped, OP_ALIGN); pool.addOperation(deletionLength, OP_DELETE); }
            else head.position += deletionLength;
            const u32 tailEndClipped = u32(candEndClipped(tail, cw));
            const u32 rightMapped = u32(i64(candObservedLength(head)) + candEndClipped(head, cw) - i64(leftMapped) - i64(tailEndClipped));
            if (rightMapped) pool.addOperation(rightMapped, OP_ALIGN);
            if (tailEndClipped) pool.addOperation(tailEndClipped, OP_SOFT_CLIP);
            const u16 tailRightClipped = tail.reverse ? tail.lowClipped : tail.highClipped;
            // resetAlignment reads the OLD cigar for the unclipped position, then points at the new one
            const i64 unclipped = candUnclippedPosition(head, cw);
            CigarPool view = pool; view.used = cigarOffset;
            candResetAlignment(head, view);
            head.position = unclipped;
            if (head.reverse) head.lowClipped = tailRightClipped; else head.highClipped = tailRightClipped;
            updateFragmentCigar(P, R, scanRead ? *scanRead : read, head, head.position + clippingPositionOffset, pool, cigarOffset);
            ++simpleIndels;
        }
    }
}

ISAAC_HD void alignSimpleInsertion(const DevParams &P, const DevReference &R, const ReadView &read, CigarPool &pool, Cand &head, u32 headSeedOffset, u32 headSeedLength,
                                   Cand &tail, u32 tailSeedOffset, u32 tailSeedLength, u32 &simpleIndels, const RefView *referenceOverride = 0, const ReadView *scanRead = 0, u32 lanes = 1, u32 lane = 0)
{
    const u32 *cw = pool.words;
    if (i64(headSeedOffset) < candBeginClipped(head, cw)) return;
    if (candBeginClipped(tail, cw) + i64(candObservedLength(tail)) < i64(tailSeedOffset + tailSeedLength)) return;
    const u32 tailOffset = headSeedOffset + headSeedLength;
    const u32 observedEnd = u32(candBeginClipped(tail, cw) + i64(candObservedLength(tail)));
    const i64 headUnclipped = candUnclippedPosition(head, cw), tailUnclipped = candUnclippedPosition(tail, cw);
    const i64 insertionLengthL = headUnclipped - tailUnclipped;
    if (insertionLengthL < 0) return;
    const u32 insertionLength = u32(insertionLengthL);
    if (tailSeedOffset - headSeedOffset < insertionLength + headSeedLength) return;   // unsigned arithmetic as in the reference
    const bool reverse = head.reverse;
    RefView reference; if (referenceOverride) reference = *referenceOverride; else { reference.p = R.bases + R.contigOffset[head.contigId]; reference.base = 0; }
    const i64 refSize = i64(contigLength(R, head.contigId));
    i64 tailIt = i64(tailOffset) + insertionLength;
    u32 tailLength = observedEnd - tailOffset - insertionLength;
    const u32 tailMismatches = countMismatches(read, reverse, tailIt, reference, headUnclipped + tailOffset, refSize, tailLength, lanes, lane);
    u32 leftFlankMismatches = countMismatches(read, reverse, tailIt - insertionLength - GAP_FLANK_BASES, reference, headUnclipped + tailOffset - GAP_FLANK_BASES, refSize, GAP_FLANK_BASES, lanes, lane);
    u32 rightFlankMismatches = countMismatches(read, reverse, tailIt, reference, headUnclipped + tailOffset, refSize, imin(GAP_FLANK_BASES, tailLength), lanes, lane);
    u32 rightRealignedMismatches = tailMismatches, leftRealignedMismatches = 0;
    i64 refIt = headUnclipped + tailOffset;
    u32 bestMismatches = tailMismatches, bestOffset = tailOffset, bestLeftFlankMismatches = leftFlankMismatches, bestRightFlankMismatches = rightFlankMismatches;
    for (u32 insertionOffset = tailOffset; bestMismatches && insertionOffset <= tailSeedOffset - insertionLength; ++insertionOffset, ++tailIt, ++refIt, --tailLength)
    {
        const u32 thisOffsetMismatches = leftRealignedMismatches + rightRealignedMismatches;
        if (bestMismatches > thisOffsetMismatches)
        {
            bestOffset = insertionOffset; bestMismatches = thisOffsetMismatches;
            bestLeftFlankMismatches = leftFlankMismatches; bestRightFlankMismatches = rightFlankMismatches;
        }
        const bool newLeftMismatch = !isMatch(strandBase(read, reverse, u32(tailIt - insertionLength)), reference[refIt]);
        leftRealignedMismatches += newLeftMismatch; leftFlankMismatches += newLeftMismatch;
        if (insertionOffset >= GAP_FLANK_BASES)
            leftFlankMismatches -= !isMatch(strandBase(read, reverse, u32(tailIt - insertionLength - GAP_FLANK_BASES)), reference[refIt - GAP_FLANK_BASES]);
        const bool disappearingRightMismatch = !isMatch(strandBase(read, reverse, u32(tailIt)), reference[refIt]);
        rightRealignedMismatches -= disappearingRightMismatch; rightFlankMismatches -= disappearingRightMismatch;
        if (tailLength > GAP_FLANK_BASES)
            rightFlankMismatches += !isMatch(strandBase(read, reverse, u32(tailIt + GAP_FLANK_BASES)), reference[refIt + GAP_FLANK_BASES]);
    }
    const i64 clippingPositionOffset = candBeginClipped(head, cw);
    const u32 leftMapped = u32(i64(bestOffset) - clippingPositionOffset);
    if (!leftMapped) return;   // ISAAC_ASSERT in the reference
    const u32 headMismatches = countMismatches(read, reverse, clippingPositionOffset, reference, head.position, refSize, leftMapped, lanes, lane);
    const u32 newMismatches = headMismatches + bestMismatches;
    const u32 sws = P.normalizedMismatchScore * newMismatches + P.normalizedGapOpenScore +
        imin(P.normalizedMaxGapExtendScore, (insertionLength - 1) * P.normalizedGapExtendScore);
    if (bestLeftFlankMismatches <= GAP_FLANK_MISMATCHES_MAX && bestRightFlankMismatches <= GAP_FLANK_MISMATCHES_MAX)
    {
        if (tail.smithWatermanScore > sws || (tail.smithWatermanScore == sws && tail.mismatchCount > newMismatches))
        {
            const u32 tailEndClipped = u32(candEndClipped(tail, cw));
            const u32 rightMapped = u32(i64(candObservedLength(head)) + candEndClipped(head, cw) - i64(leftMapped) - i64(tailEndClipped) - i64(insertionLength));
            if (!rightMapped) return; // ISAAC_ASSERT in the reference
            const u32 cigarOffset = pool.used;
            if (clippingPositionOffset) pool.addOperation(u32(clippingPositionOffset), OP_SOFT_CLIP);
            pool.addOperation(leftMapped, OP_ALIGN);
            pool.addOperation(insertionLength, OP_INSERT);
            pool.addOperation(rightMapped, OP_ALIGN);
            if (tailEndClipped) pool.addOperation(tailEndClipped, OP_SOFT_CLIP);
            const u16 headLeftClipped = head.reverse ? head.highClipped : head.lowClipped;
            const i64 unclipped = candUnclippedPosition(tail, cw);
            CigarPool view = pool; view.used = cigarOffset;
            candResetAlignment(tail, view);
            tail.position = unclipped;
            if (tail.reverse) tail.highClipped = headLeftClipped; else tail.lowClipped = headLeftClipped;
            updateFragmentCigar(P, R, scanRead ? *scanRead : read, tail, head.position, pool, cigarOffset);
            ++simpleIndels;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Sort keys of a list in fast memory (LDS in the kernels): entry i at a[i * stride], b[i * stride].  The comparators of the fragment stage
// are lexicographic orders on a few integer fields of the candidates; stated on two 64-bit words per candidate they give the same answer
// for every pair, so the same sort makes the same permutation without reading the candidates through their pointers at every comparison.
struct LeanKeyArea { u64 *a; u64 *b; u32 stride; };
ISAAC_HD bool leanKeyLess(const LeanKeyArea &k, u32 i, u32 j)
{
    const u64 ai = k.a[i * k.stride], aj = k.a[j * k.stride];
    if (ai != aj) return ai < aj;
    return k.b[i * k.stride] < k.b[j * k.stride];
}
struct KeyLessByIndex { LeanKeyArea k; ISAAC_HD bool operator()(u8 x, u8 y) const { return leanKeyLess(k, x, y); } };
// FragmentMetadata::operator< as two integers: (contig, position) and (strand, observed length).  Positions of candidates lie within
// a read length of their contig, and a contig has fewer than 2^40 bases.
static const i64 LEAN_POSITION_BIAS = 4096;
ISAAC_HD u64 leanPositionKey(u32 contigId, i64 position) { return (u64(contigId) << 41) | u64(position + LEAN_POSITION_BIAS); }

// candidate list of one read: elements never move, the list is an index array (see sort.h)
struct CandList
{
    Cand *store; u8 *order; u32 n; u32 stored; u32 capacity; u32 overflow;
    const LeanKeyArea *keys; u32 keyCap;        // optional: room for the sort keys of keyCap candidates, indexed like store
    ISAAC_HD Cand &at(u32 i) { return store[order[i]]; }
    ISAAC_HD const Cand &at(u32 i) const { return store[order[i]]; }
};

struct CandLessByPosition
{
    const Cand *store;
    ISAAC_HD bool operator()(u8 a, u8 b) const { return candLess(store[a], store[b]); }
};
// SimpleIndelAligner.cpp:443-449
struct CandLessByUnclippedPosition
{
    const Cand *store; const u32 *pool;
    ISAAC_HD bool operator()(u8 a, u8 b) const
    {
        const Cand &l = store[a], &r = store[b];
        return l.contigId < r.contigId || (l.contigId == r.contigId && candUnclippedPosition(l, pool) < candUnclippedPosition(r, pool));
    }
};

// FragmentBuilder::consolidateDuplicateFragments (FragmentBuilder.cpp:279-324)
ISAAC_HD void consolidateDuplicateFragments(CandList &l, bool removeUnaligned)
{
    if (l.keys && l.stored <= l.keyCap)
    {
        for (u32 i = 0; i < l.n; ++i)
        {
            const u32 at = l.order[i]; const Cand &c = l.store[at];
            l.keys->a[at * l.keys->stride] = leanPositionKey(c.contigId, c.position); l.keys->b[at * l.keys->stride] = (u64(c.reverse) << 32) | c.observedLength;
        }
        KeyLessByIndex less; less.k = *l.keys;
        exactSort(l.order, i32(l.n), less);
    }
    else
    {
        CandLessByPosition less; less.store = l.store;
        exactSort(l.order, i32(l.n), less);
    }
    u32 first = 0;
    while (first != l.n && removeUnaligned && !candAligned(l.at(first))) ++first;
    if (first) { for (u32 i = first; i < l.n; ++i) l.order[i - first] = l.order[i]; l.n -= first; }
    if (2 > l.n) return;
    u32 last = 0;
    for (u32 current = 1; current != l.n; ++current)
    {
        if (removeUnaligned && !candAligned(l.at(current))) { }
        else if (candEqual(l.at(last), l.at(current)))
        {
            Cand &a = l.at(last); const Cand &b = l.at(current);
            a.uniqueSeedCount = u16(a.uniqueSeedCount + b.uniqueSeedCount);
            a.nonUniqueFirst = imin(a.nonUniqueFirst, b.nonUniqueFirst);
            a.nonUniqueSecond = imax(a.nonUniqueSecond, b.nonUniqueSecond);
        }
        else { ++last; if (last != current) l.order[last] = l.order[current]; }
    }
    l.n = last + 1;
}

// SimpleIndelAligner::alignSimpleIndels (SimpleIndelAligner.cpp:460-518)
// SimpleIndelAligner::alignSimpleIndels (SimpleIndelAligner.cpp:460-518) in three parts, so that the expensive middle one can
// run in a kernel of its own over just the reads that need it: sort by unclipped position, "is there a pair to look at",
// the pairs.
ISAAC_HD void sortForSimpleIndels(CandList &l, const CigarPool &pool)
{
    if (l.keys && l.stored <= l.keyCap)
    {
        for (u32 i = 0; i < l.n; ++i)
        {
            const u32 at = l.order[i]; const Cand &c = l.store[at];
            l.keys->a[at * l.keys->stride] = leanPositionKey(c.contigId, candUnclippedPosition(c, pool.words)); l.keys->b[at * l.keys->stride] = 0;
        }
        KeyLessByIndex less; less.k = *l.keys;
        exactSort(l.order, i32(l.n), less);
        return;
    }
    CandLessByUnclippedPosition less; less.store = l.store; less.pool = pool.words;
    exactSort(l.order, i32(l.n), less);
}
ISAAC_HD bool simpleIndelPairQualifies(const DevParams &P, const Cand &head, const Cand &tail, const u32 *pool)
{
    if (head.contigId != tail.contigId || head.reverse != tail.reverse) return false;
    const i64 distance = candUnclippedPosition(tail, pool) - candUnclippedPosition(head, pool);
    if (!distance) return false;  // ISAAC_ASSERT in the reference
    return (distance < 0 ? -distance : distance) < i64(P.semialignedGapLimit);
}
ISAAC_HD bool hasSimpleIndelPair(const DevParams &P, const CandList &l, const CigarPool &pool)
{
    for (u32 t = 1; t < l.n; ++t) if (simpleIndelPairQualifies(P, l.at(t - 1), l.at(t), pool.words)) return true;
    return false;
}
// LDS staging area of the wave-per-cluster form: the read's BCL bytes and a window of the contig around a candidate pair
struct IndelStage { u8 *bcl; u32 bclCap; char *window; u32 windowCap; u32 lane; };
static const i64 INDEL_WINDOW_MARGIN = 256;   // the detector looks at most gap limit + read length + 2 flanks around the candidates

ISAAC_HD void simpleIndelPairs(const DevParams &P, const DevReference &R, const ReadView &readIn, CigarPool &pool, CandList &l, u32 &simpleIndels, const IndelStage *stage = 0)
{
    ReadView read = readIn;
#if defined(__HIP_DEVICE_COMPILE__)
    if (stage && read.length <= stage->bclCap)
    {
        __syncthreads();                                           // nobody still reads the previous read's copy
        for (u32 i = stage->lane; i < read.length; i += 64) stage->bcl[i] = readIn.bcl[i];
        __syncthreads();
        read.bcl = stage->bcl;
    }
#endif
    for (u32 t = 1; t != l.n; ++t)
    {
        Cand &head = l.at(t - 1); Cand &tail = l.at(t);
        if (simpleIndelPairQualifies(P, head, tail, pool.words))
        {
            const RefView *window = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            RefView windowView;
            if (stage)
            {
                const i64 hu = candUnclippedPosition(head, pool.words), tu = candUnclippedPosition(tail, pool.words);
                const i64 lo = imin(hu, tu) - INDEL_WINDOW_MARGIN, hi = imax(hu, tu) + i64(read.length) + INDEL_WINDOW_MARGIN;
                if (hi - lo <= i64(stage->windowCap))
                {
                    const char *contig = R.bases + R.contigOffset[head.contigId];
                    const char *first = R.bases, *last = R.bases + R.totalBases;
                    __syncthreads();                                   // the previous pair's readers are done with the window
                    for (i64 i = stage->lane; i < hi - lo; i += 64) { const char *g = contig + lo + i; stage->window[i] = (g >= first && g < last) ? *g : char(0); }
                    __syncthreads();
                    windowView.p = stage->window; windowView.base = lo; window = &windowView;
                }
            }
#endif
            const DevSeed &headSeed = P.seeds[head.firstSeedIndex]; const DevSeed &tailSeed = P.seeds[tail.firstSeedIndex];
            const i64 readLength = read.length;
            const i64 headSeedOffset = head.reverse ? readLength - headSeed.offset - headSeed.length : i64(headSeed.offset);
            const i64 tailSeedOffset = head.reverse ? readLength - tailSeed.offset - tailSeed.length : i64(tailSeed.offset);
            if (0 < tailSeedOffset - headSeedOffset)
                alignSimpleDeletion(P, R, read, pool, head, u32(headSeedOffset), tail, u32(tailSeedOffset), tailSeed.length, simpleIndels, window, &readIn, stage ? 64u : 1u, stage ? stage->lane : 0u);
            else
                alignSimpleInsertion(P, R, read, pool, tail, u32(tailSeedOffset), tailSeed.length, head, u32(headSeedOffset), headSeed.length, simpleIndels, window, &readIn, stage ? 64u : 1u, stage ? stage->lane : 0u);
        }
    }
}
ISAAC_HD void alignSimpleIndels(const DevParams &P, const DevReference &R, const ReadView &read, CigarPool &pool, CandList &l, u32 &simpleIndels)
{
    if (l.n < 2) return;
    sortForSimpleIndels(l, pool);
    simpleIndelPairs(P, R, read, pool, l, simpleIndels);
}

// One gapped (banded Smith-Waterman) re-alignment problem: GappedAligner::alignGapped of a candidate.  Problems are collected
// per chunk and executed 16 lanes per problem (k_gapped_jobs); results come back to the cluster's thread.
// accepted: set by the rescue's sums for the retries whose result replaces the ungapped shadow (sums.h); adapterRange: the strand's sequencing adapter (0: none)
struct GappedJob { Cand in; u32 cluster; u16 endCyclesMasked, accepted; u32 tag; u32 adapterRange; };
struct GappedResult { Cand out; u32 matchCount; u32 nCigar; u32 cigar[40]; };
static_assert(sizeof(GappedJob) == 80 && sizeof(GappedResult) == 232, "gapped job layouts");

// per-thread scratch of the fragment stage's general form: where the index arrays live is the caller's business (LDS in the kernels)
struct FragmentWork
{
    u8 *order;                    // CAND_CAP entries
    u8 *matchOrder;               // MATCH_CAP_MAX entries (not needed when every cluster's matches fit the MatchStage)
    u32 *tflags;                  // 3 * 512 words: banded SW traceback flags, reads up to 512 cycles (serial gapped alignment only)
    const LeanKeyArea *keys; u32 keyCap;   // optional room for the sort keys of keyCap candidates (CandList::keys)
};
struct FragmentWorkStore
{
    u8 order[CAND_CAP]; u8 matchOrder[MATCH_CAP_MAX]; u32 tflags[3 * 512];
    FragmentWork bind() { FragmentWork w; w.order = order; w.matchOrder = matchOrder; w.tflags = tflags; w.keys = 0; w.keyCap = 0; return w; }
};

struct MatchLess
{
    const Match *m;
    // SelectMatchesTransition.cpp:242-254 within one cluster: (location, seed index); the reverse bit makes the order total
    ISAAC_HD bool operator()(u8 a, u8 b) const
    {
        const Match &l = m[a], &r = m[b];
        if (l.location != r.location) return l.location < r.location;
        const u32 ls = seedIdSeed(l.seedId), rs = seedIdSeed(r.seedId);
        if (ls != rs) return ls < rs;
        return (l.seedId & 1) < (r.seedId & 1);
    }
};

// FragmentBuilder::build (FragmentBuilder.cpp:82-145) + alignFragments (:147-217) for one cluster.
// `matches` are the cluster's Match records in any order.  Results go to `out` (lists compacted in final order).
// the list order l.order applied to l.store in place (cycle following); the first l.n elements end up in list order
ISAAC_HD void applyOrderInPlace(Cand *store, u8 *order, u32 n)
{
    for (u32 i = 0; i < n; ++i)
    {
        u32 src = order[i];
        while (src < i) src = order[src];       // already moved: follow where it went
        if (src != i) { const Cand t = store[i]; store[i] = store[src]; store[src] = t; }
        order[i] = u8(src);
    }
}

// FragmentBuilder::build (FragmentBuilder.cpp:82-145) + alignFragments (:147-217) in three steps, so that the ungapped scans can
// run one per thread (k_align_candidates) instead of one cluster's worth per thread:
//   buildCandidates   matches -> candidate positions of both reads, duplicates merged (list order in out.cands)
//   alignCandidate    UngappedAligner::alignUngapped for one candidate; its CIGAR takes a fixed 3-word slot of the cluster's pool
//   finishCandidates  consolidation, the single-indel stage (or its deferral), lists compacted in place
// The matches of a cluster with few of them, copied once to fast memory (k_build_fragments: LDS): location, seed and strand are all
// the fragment stage reads of a match, and it reads them many times -- the counts, every comparison of the sort, the two passes that
// make the candidates -- each time through a pointer into global memory.  keys / ties are interleaved by `stride` (entry i of this
// thread at [i * stride]), order is this thread's own bytes.
struct MatchStage { u64 *keys; u8 *ties; u8 *order; u32 stride; u32 cap; };
struct StagedMatchLess
{
    const u64 *keys; const u8 *ties; u32 stride;
    ISAAC_HD bool operator()(u8 a, u8 b) const
    {
        const u64 l = keys[a * stride], r = keys[b * stride];
        if (l != r) return l < r;
        return ties[a * stride] < ties[b * stride];         // (seed index, strand), as MatchLess
    }
};
ISAAC_HD bool buildCandidates(const DevParams &P, const u8 *clusterBcl, const Match *matches, u32 nMatches, bool trim, FragmentWork &work, ClusterFragments &out,
                              const MatchStage *stage = 0)
{
    out.nCands[0] = out.nCands[1] = 0; out.cigarUsed = 0; out.flags = 0; out.repeatSeedsCount = 0; out.built = 0;
    out.cands[1] = out.cands[0]; out.candCap[1] = out.candCap[0];
    STAMP_BEGIN();
    for (u32 r = 0; r < 2; ++r)
        out.endCyclesMasked[r] = (trim && r < P.nReads) ? trimLowQualityEnd(clusterBcl + P.readOffset[r], P.readLength[r], P.baseQualityCutoff) : 0;
    STAMP(20);
    if (!nMatches || nMatches > MATCH_CAP_MAX) { if (nMatches > MATCH_CAP_MAX) out.flags |= CLUSTER_OVERFLOW; return false; }
    // seedMatchCounts_ / repeatSeedsCount_ (FragmentBuilder.cpp:99-126): order independent once stated per seed index
    u32 counts[MAX_SEEDS]; bool tooMany[MAX_SEEDS];
    for (u32 s = 0; s < MAX_SEEDS; ++s) { counts[s] = 0; tooMany[s] = false; }
    bool any = false;
    const bool staged = stage && nMatches <= stage->cap;
    for (u32 i = 0; i < nMatches; ++i)
    {
        const u64 location = matches[i].location, seedId = matches[i].seedId;
        if (staged) { stage->keys[i * stage->stride] = location; stage->ties[i * stage->stride] = u8((seedIdSeed(seedId) << 1) | (seedId & 1)); }
        if (refposIsNoMatch(location)) continue;
        any = true;
        const u32 s = seedIdSeed(seedId);
        if (refposIsTooMany(location)) tooMany[s] = true; else ++counts[s];
    }
    if (!any) return false;
    u32 repeatSeedsCount = 0;
    for (u32 s = 0; s < P.nSeeds; ++s) if (tooMany[s] || counts[s] >= P.repeatThreshold) ++repeatSeedsCount;
    out.repeatSeedsCount = repeatSeedsCount;
    STAMP(21);
    // the reference adds candidates in sorted match order; that order is the input order of the first std::sort
    u8 *matchOrder = staged ? stage->order : work.matchOrder;
    for (u32 i = 0; i < nMatches; ++i) matchOrder[i] = u8(i);
    if (staged) { StagedMatchLess ml; ml.keys = stage->keys; ml.ties = stage->ties; ml.stride = stage->stride; exactSort(matchOrder, i32(nMatches), ml); }
    else { MatchLess ml; ml.m = matches; exactSort(matchOrder, i32(nMatches), ml); }
    STAMP(22);
    bool built = false;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        // the candidates are made where they stay: read 0's from the cluster's first slot, read 1's behind read 0's consolidated list
        if (r) { out.cands[1] = out.cands[0] + out.nCands[0]; out.candCap[1] = out.candCap[0] - out.nCands[0]; }
        CandList l; l.store = out.cands[r]; l.order = work.order; l.n = 0; l.stored = 0; l.capacity = imin(CAND_CAP, out.candCap[r]); l.overflow = 0; l.keys = work.keys; l.keyCap = work.keyCap;
        for (u32 k = 0; k < nMatches; ++k)
        {
            const u32 at = matchOrder[k];
            Match m;
            if (staged) { m.location = stage->keys[at * stage->stride]; m.seedId = u64(stage->ties[at * stage->stride]); }    // seed index and strand: all that is read of seedId below
            else m = matches[at];
            if (refposIsNoMatch(m.location) || refposIsTooMany(m.location)) continue;
            const u32 s = seedIdSeed(m.seedId);
            const DevSeed &seed = P.seeds[s];
            if (seed.readIndex != r || tooMany[s] || counts[s] >= P.repeatThreshold) continue;
            if (l.stored == l.capacity) { l.overflow = 1; break; }
            // FragmentBuilder::addMatch (:219-249) + getReadPosition (:326-343)
            Cand &f = l.store[l.stored];
            candInit(f, r);
            const bool reverse = m.seedId & 1;
            const i64 seedPosition = i64(refposPosition(m.location));
            f.firstSeedIndex = (signed char)s;
            f.contigId = refposContig(m.location);
            f.position = reverse ? seedPosition + seed.length + seed.offset - i64(P.readLength[r]) : seedPosition - seed.offset;
            f.reverse = reverse;
            f.repeatSeedsCount = u16(repeatSeedsCount);
            if (seed.length != 64 && (m.location & 1)) { f.nonUniqueFirst = seed.offset; f.nonUniqueSecond = seed.offset; }
            else f.uniqueSeedCount = 1;
            l.order[l.n++] = u8(l.stored++);
        }
        if (l.overflow) out.flags |= CLUSTER_OVERFLOW;
        STAMP(23);
        if (!l.n) continue;
        built = true;
        consolidateDuplicateFragments(l, false);       // alignFragments (:147-217) starts with this
        STAMP(24);
        applyOrderInPlace(out.cands[r], work.order, l.n);
        out.nCands[r] = l.n;
    }
    out.built = built;
    return built;
}

// the four adapter ranges (read x strand) of the cluster whose view this is; NULL without adapters
ISAAC_HD u32 *clusterAdapterRanges(const DevParams &P, const ClusterFragments &f) { return P.adapters ? P.adapterRanges + 4 * size_t(f.cands[0] - P.adapterCandBase) : (u32 *)0; }
// FragmentSequencingAdapterClipper::checkInitStrand as FragmentBuilder::alignFragments calls it (FragmentBuilder.cpp:164-174): per read a fresh clipper, the
// first candidate of either strand in list order (the list is consolidated: sorted) decides that strand's range.  Runs on the candidates as built, before
// any of them is aligned (alignUngapped moves positions).
ISAAC_HD void clusterInitAdapterRanges(const DevParams &P, const DevReference &R, const u8 *clusterBcl, const ClusterFragments &f, u32 r, u32 strand)
{
    u32 *ranges = clusterAdapterRanges(P, f);
    if (!ranges) return;
    u32 range = 0;
    const Cand *list = f.list(r);
    const u32 n = f.listLength(r);
    for (u32 i = 0; i < n; ++i)
    {
        if (u32(0 != list[i].reverse) != strand) continue;
        ReadView read; read.bcl = clusterBcl + P.readOffset[r]; read.length = P.readLength[r]; read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = 0;
        range = adapterStrandRange(*P.adapters, R, read, 0 != strand, list[i].contigId, list[i].position);
        break;
    }
    ranges[2 * r + strand] = range;
}
ISAAC_HD void alignCandidate(const DevParams &P, const DevReference &R, const u8 *clusterBcl, ClusterFragments &out, u32 r, u32 i, Counters &cnt)
{
    const u32 *ranges = clusterAdapterRanges(P, out);
    ReadView read; read.bcl = clusterBcl + P.readOffset[r]; read.length = P.readLength[r]; read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = r ? out.endCyclesMasked[1] : out.endCyclesMasked[0];      // (r is known at run time only: selects, not indexes -- ClusterFragments::list)
    const u32 slot = 3 * ((r ? out.nCands[0] : 0) + i);
    CigarPool pool; pool.words = out.cigarPool; pool.used = slot; pool.capacity = slot + 3; pool.overflow = 0;
    // (the candidate in registers while the scan fills it in, one load and one store: updated in its place it went to device memory more than once)
#if defined(ISAAC_CAND_IN_PLACE)       // (the form of rounds 1-5, for comparison)
    alignUngapped(P, R, read, out.list(r)[i], pool);
#else
    Cand c = out.list(r)[i];
    alignUngapped(P, R, read, c, pool, ranges ? ranges[2 * r + (c.reverse ? 1 : 0)] : 0u);
    out.list(r)[i] = c;
#endif
    ++cnt.ungappedScans;
}

ISAAC_HD void finishCandidates(const DevParams &P, const DevReference &R, const u8 *clusterBcl, FragmentWork &work, ClusterFragments &out, Counters &cnt, bool deferSimpleIndels)
{
    if (!out.built) return;
    STAMP_BEGIN();
    CigarPool pool; pool.words = out.cigarPool; pool.used = 3 * (out.nCands[0] + out.nCands[1]); pool.capacity = out.cigarCap; pool.overflow = 0;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        const u32 n = out.nCands[r];
        if (!n) continue;
        ReadView read; read.bcl = clusterBcl + P.readOffset[r]; read.length = P.readLength[r]; read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = out.endCyclesMasked[r];
        CandList l; l.store = out.cands[r]; l.order = work.order; l.n = n; l.stored = n; l.capacity = CAND_CAP; l.overflow = 0; l.keys = work.keys; l.keyCap = work.keyCap;
        for (u32 i = 0; i < n; ++i) work.order[i] = u8(i);
        consolidateDuplicateFragments(l, true);
        STAMP(26);
        if (P.semialignedGapLimit)
        {
            if (l.n >= 2)
            {
                sortForSimpleIndels(l, pool);
                if (hasSimpleIndelPair(P, l, pool))
                {
                    if (deferSimpleIndels) out.flags |= CLUSTER_INDEL_PENDING << r;   // the list is left in this order for finishSimpleIndels
                    else { u32 si = 0; simpleIndelPairs(P, R, read, pool, l, si); cnt.simpleIndels += si; }
                }
            }
            if (!(out.flags & (CLUSTER_INDEL_PENDING << r))) consolidateDuplicateFragments(l, true);
        }
        STAMP(27);
        // the candidates stay in list order; the gapped retries (FragmentBuilder.cpp:187-214) follow in finishFragments
        applyOrderInPlace(out.cands[r], work.order, l.n);
        out.nCands[r] = l.n;
        STAMP(28);
    }
    out.cigarUsed = pool.used;
    if (pool.overflow) out.flags |= CLUSTER_OVERFLOW;
}

// the three steps one after the other in this thread (serial form)
ISAAC_HD bool buildFragments(const DevParams &P, const DevReference &R, const u8 *clusterBcl, const Match *matches, u32 nMatches,
                             bool withGaps, bool trim, FragmentWork &work, ClusterFragments &out, Counters &cnt, bool deferSimpleIndels = false)
{
    (void)withGaps;
    if (!buildCandidates(P, clusterBcl, matches, nMatches, trim, work, out)) return false;
    if (P.adapters) for (u32 r = 0; r < P.nReads; ++r) for (u32 strand = 0; strand < 2; ++strand) clusterInitAdapterRanges(P, R, clusterBcl, out, r, strand);
    for (u32 r = 0; r < P.nReads; ++r) for (u32 i = 0; i < out.nCands[r]; ++i) alignCandidate(P, R, clusterBcl, out, r, i, cnt);
    finishCandidates(P, R, clusterBcl, work, out, cnt, deferSimpleIndels);
    return true;
}

// The single-indel stage of the reads buildFragments left pending: the pair loop of alignSimpleIndels and the consolidation
// that follows it (FragmentBuilder.cpp:176-185).  The list was stored sorted by unclipped position.
ISAAC_HD void finishSimpleIndels(const DevParams &P, const DevReference &R, const u8 *clusterBcl, ClusterFragments &out, u8 *order, Counters &cnt, const IndelStage *stage = 0)
{
    CigarPool pool; pool.words = out.cigarPool; pool.used = out.cigarUsed; pool.capacity = out.cigarCap; pool.overflow = 0;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        if (!(out.flags & (CLUSTER_INDEL_PENDING << r))) continue;
        out.flags &= ~(CLUSTER_INDEL_PENDING << r);
        ReadView read; read.bcl = clusterBcl + P.readOffset[r]; read.length = P.readLength[r]; read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = out.endCyclesMasked[r];
        const u32 n = out.nCands[r];
        CandList l; l.store = out.cands[r]; l.order = order; l.n = n; l.stored = n; l.capacity = CAND_CAP; l.overflow = 0; l.keys = 0; l.keyCap = 0;
        for (u32 i = 0; i < n; ++i) order[i] = u8(i);
        u32 si = 0;
        simpleIndelPairs(P, R, read, pool, l, si, stage);
        cnt.simpleIndels += si;
        consolidateDuplicateFragments(l, true);
        for (u32 i = 0; i < l.n; ++i)      // the permutation in place (cycle following), as in finishFragments
        {
            u32 src = order[i];
            while (src < i) src = order[src];
            if (src != i) { const Cand t = out.cands[r][i]; out.cands[r][i] = out.cands[r][src]; out.cands[r][src] = t; }
            order[i] = u8(src);
        }
        out.nCands[r] = l.n;
    }
    out.cigarUsed = pool.used;
    if (pool.overflow) out.flags |= CLUSTER_OVERFLOW;
}

// the candidates FragmentBuilder::alignFragments would hand to GappedAligner (FragmentBuilder.cpp:197): list order, read 0 first
ISAAC_HD u32 countGappedJobs(const ClusterFragments &f, bool withGaps)
{
    u32 n = 0;
    if (withGaps && f.built) for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) n += BSW_MISMATCHES_CUTOFF < f.cands[r][i].mismatchCount;
    return n;
}
ISAAC_HD void makeGappedJob(const DevParams &P, const ClusterFragments &f, u32 r, u32 i, u32 chunkCluster, GappedJob &j)
{
    const u32 *ranges = clusterAdapterRanges(P, f);
    j.in = f.cands[r][i]; j.cluster = chunkCluster; j.endCyclesMasked = u16(f.endCyclesMasked[r]); j.accepted = 0; j.tag = (r << 16) | i;
    j.adapterRange = ranges ? ranges[2 * r + (j.in.reverse ? 1 : 0)] : 0u;
    // FragmentMetadata::resetAlignment starts from the unclipped position; the job does not carry the old CIGAR
    j.in.position = candUnclippedPosition(f.cands[r][i], f.cigarPool); j.in.cigarLength = 0; j.in.cigarOffset = 0;
}
ISAAC_HD void writeGappedJobs(const DevParams &P, const ClusterFragments &f, u32 chunkCluster, GappedJob *jobs)
{
    u32 n = 0;
    for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) if (BSW_MISMATCHES_CUTOFF < f.cands[r][i].mismatchCount)
        makeGappedJob(P, f, r, i, chunkCluster, jobs[n++]);
}

// one gapped problem in the calling thread (serial form of k_gapped_jobs)
ISAAC_HD void runGappedJobSerial(const DevParams &P, const DevReference &R, const u8 *clusterBcl, const GappedJob &job, u32 *tflags, GappedResult &res)
{
    ReadView read; const u32 r = job.in.readIndex;
    read.bcl = clusterBcl + P.readOffset[r]; read.length = P.readLength[r]; read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = job.endCyclesMasked;
    res.out = job.in;
    CigarPool pool; pool.words = res.cigar; pool.used = 0; pool.capacity = 40; pool.overflow = 0;
    res.matchCount = alignGapped(P, R, read, res.out, pool, tflags, job.adapterRange);
    res.nCigar = pool.overflow ? 0xffffffffu : pool.used;
}

// second half of FragmentBuilder::alignFragments (FragmentBuilder.cpp:187-216): accept rule for the gapped retries, then the
// final consolidation.
// `provider(r, i)` returns the GappedResult of candidate i of read r (called in countGappedJobs order) or NULL for "no gapped alignment"
template <typename ProviderF>
ISAAC_HD void finishFragments(const DevParams &P, ClusterFragments &out, ProviderF &provider, u8 *order, Counters &cnt, const LeanKeyArea *keys = 0, u32 keyCap = 0)
{
    if (!out.built) return;
    CigarPool pool; pool.words = out.cigarPool; pool.used = out.cigarUsed; pool.capacity = out.cigarCap; pool.overflow = 0;
    for (u32 r = 0; r < P.nReads; ++r)
    {
        const u32 n = out.nCands[r];
        for (u32 i = 0; i < n; ++i)
        {
            Cand &f = out.cands[r][i];
            const GappedResult *pg = (BSW_MISMATCHES_CUTOFF < f.mismatchCount) ? provider(r, i) : 0;
            if (pg)
            {
                const GappedResult &g = *pg;
                ++cnt.bswJobs;
                if (0xffffffffu == g.nCigar) { out.flags |= CLUSTER_OVERFLOW; continue; }
                const Cand &tmp = g.out;
                if (g.matchCount && g.matchCount + BSW_WIDEST_GAP_SIZE > candObservedLength(f) && (tmp.mismatchCount <= P.gappedMismatchesMax) &&
                    (f.mismatchCount > tmp.mismatchCount) && lpLess(f.logProbability, tmp.logProbability))
                {
                    f = tmp;
                    f.cigarOffset = pool.used;
                    for (u32 w = 0; w < g.nCigar; ++w) pool.push(g.cigar[w]);
                    ++cnt.bswAccepted;
                }
            }
        }
        CandList l; l.store = out.cands[r]; l.order = order; l.n = n; l.stored = n; l.capacity = CAND_CAP; l.overflow = 0; l.keys = keys; l.keyCap = keyCap;
        for (u32 i = 0; i < n; ++i) order[i] = u8(i);
        consolidateDuplicateFragments(l, true);
        // apply the permutation in place (cycle following); order[i] = index of the element that belongs at i
        for (u32 i = 0; i < l.n; ++i)
        {
            u32 src = order[i];
            while (src < i) src = order[src];       // already moved: follow where it went
            if (src != i) { const Cand t = out.cands[r][i]; out.cands[r][i] = out.cands[r][src]; out.cands[r][src] = t; }
            order[i] = u8(src);
        }
        out.nCands[r] = l.n;
        cnt.candidates += l.n;
    }
    out.cigarUsed = pool.used;
    if (pool.overflow) out.flags |= CLUSTER_OVERFLOW;
}


} // namespace isaac
