// BAM alignment records on the device: what build::Build serialises per bin with bam::serializeAlignment over
// build::FragmentAccessorBamAdapter (include/bam/Bam.hh:257-345, include/build/FragmentAccessorBamAdapter.hh:62-377), in the order of
// PackedFragmentBuffer::orderForBam (include/build/PackedFragmentBuffer.hh:149-176), for --realign-gaps no --mark-duplicates 0
// --bam-exclude-tags ZX,ZY (the default tag set: SM AS RG NM BC).
//
// The records of all tiles of a call are ordered together: 128-bit keys (position | global cluster id, unmapped, second read) sorted by
// two stable radix passes; record sizes are scanned in that order; every record is then written where it belongs, one thread each.
#pragma once
#include "cluster_ops.h"

namespace isaac
{

// one tile's worth of isaac_gpu_select output plus what names its reads
struct BamTile
{
    const u8 *bcl; const FragmentRecord *records; const u32 *cigars; u64 firstRecord;   // index of the tile's first record among all records of the call
    u32 nRecords, nameLength; char name[64];                                             // "<flowcell>:<lane>:<tile>:" (FragmentAccessorBamAdapter::readName)
};
struct BamOptions { u32 nReads, readLength[2], readOffset[2], clusterLength, forcedDodgyAlignmentScore, pessimisticMapQ, barcodeLength, readGroupLength; char barcode[64], readGroup[64]; };

static const u64 INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE = 1000000000ull;   // include/build/FragmentIndex.hh:33
static const u16 DODGY_ALIGNMENT_SCORE = 0xffff;                               // io::FragmentHeader::DODGY_ALIGNMENT_SCORE

ISAAC_HD u32 decimalDigits(u32 v) { u32 n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
// Bam.hh:237-246
ISAAC_HD u32 bamReg2bin(u32 beg, u32 end)
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}
// FragmentAccessorBamAdapter::mapq (:250-265)
ISAAC_HD u32 bamMapq(const FragmentRecord &r, const BamOptions &o)
{
    if (r.flags & 256)
    {
        if (DODGY_ALIGNMENT_SCORE == r.templateAlignmentScore) return o.forcedDodgyAlignmentScore;
        return imin<u32>(60u, o.pessimisticMapQ ? imin<u32>(r.alignmentScore, r.templateAlignmentScore) : imax<u32>(r.alignmentScore, r.templateAlignmentScore));
    }
    return DODGY_ALIGNMENT_SCORE == r.alignmentScore ? o.forcedDodgyAlignmentScore : imin<u32>(60u, r.alignmentScore);
}
// FragmentAccessorBamAdapter::flag (:347-362); record flags: bit0 paired,1 unmapped,2 mateUnmapped,3 reverse,4 mateReverse,5 first,6 second,7 failFilter,8 properPair
ISAAC_HD u32 bamFlag(const FragmentRecord &r)
{
    const u32 f = r.flags; const bool paired = f & 1;
    return (paired ? 1u : 0u) | ((f & 256) ? 2u : 0u) | ((f & 2) ? 4u : 0u) | ((paired && (f & 4)) ? 8u : 0u) | ((f & 8) ? 16u : 0u) | ((f & 16) ? 32u : 0u) |
           ((paired && (f & 32)) ? 64u : 0u) | ((paired && (f & 64)) ? 128u : 0u) | ((f & 128) ? 512u : 0u);
}
ISAAC_HD bool bamStored(const FragmentRecord &r) { return 0 == (r.reserved & RECORD_NOT_STORED); }
// is the record part of the unaligned bin (both reads of the template unplaced)?  Shadows travel with their singleton.
ISAAC_HD bool bamUnalignedBin(const FragmentRecord &r) { return refposIsNoMatch(r.fStrandPosition); }

ISAAC_HD u32 bamReadNameLength(const BamTile &t, const FragmentRecord &r) { return t.nameLength + decimalDigits(r.clusterId) + 2; }   // + ":0"
ISAAC_HD u32 bamRecordBytes(const BamTile &t, const FragmentRecord &r, const BamOptions &o)
{
    const bool aligned = !(r.flags & 2);
    u32 n = 4 + 32 + bamReadNameLength(t, r) + 1 + (aligned ? 4 * u32(r.cigarLength) : 0) + (r.readLength + 1) / 2 + r.readLength;
    if (DODGY_ALIGNMENT_SCORE != r.alignmentScore) n += 7;                                                    // SM:i
    if ((r.flags & 256) && DODGY_ALIGNMENT_SCORE != r.templateAlignmentScore) n += 7;                        // AS:i
    n += 3 + o.readGroupLength + 1;                                                                            // RG:Z
    n += 7;                                                                                                    // NM:i
    n += 3 + o.barcodeLength + 1;                                                                              // BC:Z
    return n;
}

struct BamWriter
{
    u8 *p;
    ISAAC_HD void u32le(u32 v) { p[0] = u8(v); p[1] = u8(v >> 8); p[2] = u8(v >> 16); p[3] = u8(v >> 24); p += 4; }
    ISAAC_HD void byte(u8 v) { *p++ = v; }
    ISAAC_HD void iTag(char a, char b, u32 v) { byte(u8(a)); byte(u8(b)); byte(u8('i')); u32le(v); }
    ISAAC_HD void zTag(char a, char b, const char *s, u32 n) { byte(u8(a)); byte(u8(b)); byte(u8('Z')); for (u32 i = 0; i < n; ++i) byte(u8(s[i])); byte(0); }
};

// bam::serializeAlignment (Bam.hh:257-345) for one record; `out` has bamRecordBytes() bytes
ISAAC_HD void bamWriteRecord(const BamTile &t, const FragmentRecord &r, const BamOptions &o, u8 *out)
{
    BamWriter w; w.p = out;
    const bool aligned = !(r.flags & 2), unalignedBin = bamUnalignedBin(r), paired = r.flags & 1;
    // FragmentAccessorBamAdapter::operator(): aligned fragments and shadows carry the bin index position, unaligned templates NoMatch
    const i32 refId = unalignedBin ? -1 : i32(refposContig(r.fStrandPosition)), pos = unalignedBin ? -1 : i32(refposPosition(r.fStrandPosition));
    const u32 nameLength = bamReadNameLength(t, r);
    const u32 nCigar = aligned ? r.cigarLength : 0;
    const u32 observed = r.observedLength;
    w.u32le(bamRecordBytes(t, r, o) - 4);
    w.u32le(u32(refId)); w.u32le(u32(pos));
    w.u32le((bamReg2bin(u32(pos), u32(pos) + (observed ? observed : 1)) << 16) | (bamMapq(r, o) << 8) | (nameLength + 1));
    w.u32le((bamFlag(r) << 16) | (nCigar & 0xffff));
    w.u32le(r.readLength);
    const bool noMate = !paired || ((r.flags & 2) && (r.flags & 4));
    w.u32le(noMate ? u32(-1) : u32(refposContig(r.mateFStrandPosition)));
    w.u32le(noMate ? u32(-1) : u32(refposPosition(r.mateFStrandPosition)));
    w.u32le(u32(r.bamTlen));
    for (u32 i = 0; i < t.nameLength; ++i) w.byte(u8(t.name[i]));
    { u32 digits = decimalDigits(r.clusterId), v = r.clusterId; for (u32 i = 0; i < digits; ++i) { w.p[digits - 1 - i] = u8('0' + v % 10); v /= 10; } w.p += digits; }
    w.byte(u8(':')); w.byte(u8('0')); w.byte(0);
    const u32 *cigar = t.cigars + r.cigarOffset;
    for (u32 i = 0; i < nCigar; ++i) w.u32le(cigar[i]);
    // the bases as FragmentCollector::storeBclAndCigar keeps them (reverse-complemented for reverse alignments, :84-96), 4 bits each
    // (bamBaseFromBclByte :218-221), then the qualities (bamQualFromBclByte :228-230)
    const u32 readIndex = (r.flags & 64) && paired ? 1u : 0u, L = r.readLength;
    const u8 *bcl = t.bcl + u64(r.clusterId) * o.clusterLength + o.readOffset[readIndex];
    const bool reverse = r.flags & 8;
    u32 packed = 0;
    for (u32 i = 0; i < L; ++i)
    {
        const u8 b = reverse ? bcl[L - 1 - i] : bcl[i];
        const u32 base = (b & 0xfc) ? 1u << (reverse ? 3u - (b & 3u) : (b & 3u)) : 15u;
        if (i & 1) w.byte(u8((packed << 4) | base)); else packed = base;
    }
    if (L & 1) w.byte(u8(packed << 4));
    for (u32 i = 0; i < L; ++i) { const u8 b = reverse ? bcl[L - 1 - i] : bcl[i]; w.byte((b & 0xfc) ? u8(b >> 2) : u8(0)); }
    if (DODGY_ALIGNMENT_SCORE != r.alignmentScore) w.iTag('S', 'M', r.alignmentScore);
    if ((r.flags & 256) && DODGY_ALIGNMENT_SCORE != r.templateAlignmentScore) w.iTag('A', 'S', r.templateAlignmentScore);
    w.zTag('R', 'G', o.readGroup, o.readGroupLength);
    w.iTag('N', 'M', r.editDistance);
    w.zTag('B', 'C', o.barcode, o.barcodeLength);
}

#if defined(__HIPCC__)
// tile of global record index i (tiles are few: linear search over firstRecord)
__device__ inline u32 bamTileOf(const BamTile *tiles, u32 nTiles, u64 i) { u32 t = 0; while (t + 1 < nTiles && tiles[t + 1].firstRecord <= i) ++t; return t; }

// orderForBam as a 128-bit key: hi = the bin index position (unaligned templates and dropped records last), lo = global cluster id,
// unmapped, second read
__global__ void k_bam_keys(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, u64 *keyHi, u64 *keyLo, u32 *index, u32 *bytes)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = tiles[t].records[i - tiles[t].firstRecord];
    const bool stored = bamStored(r);
    keyHi[i] = !stored ? ~u64(0) : bamUnalignedBin(r) ? ~u64(0) - 1 : r.fStrandPosition;
    keyLo[i] = ((u64(r.tile) * INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE + r.clusterId) << 2) | ((r.flags & 2) ? 2u : 0u) | ((r.flags & 64) ? 1u : 0u);
    index[i] = u32(i);
    bytes[i] = stored ? bamRecordBytes(tiles[t], r, o) : 0;
}
__global__ void k_bam_gather_hi(const u64 *keyHi, const u32 *order, u64 n, u64 *out) { const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) out[i] = keyHi[order[i]]; }
__global__ void k_bam_gather_bytes(const u32 *bytes, const u32 *order, u64 n, u64 *out) { const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) out[i] = bytes[order[i]]; }
// bounds[0]: the first record (in file order) of the unaligned bin; bounds[1]: the number of records written (dropped templates sort last)
__global__ void k_bam_bounds(const u64 *sortedHi, u64 n, u64 *bounds)
{
    const u64 k = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const u64 UNALIGNED = ~u64(0) - 1, DROPPED = ~u64(0), hi = sortedHi[k], before = k ? sortedHi[k - 1] : 0;
    if (hi >= UNALIGNED && (0 == k || before < UNALIGNED)) bounds[0] = k;
    if (hi == DROPPED && (0 == k || before < DROPPED)) bounds[1] = k;
}
__global__ void k_bam_encode(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, const u32 *order, const u64 *offsets, u8 *out, u64 capacity)
{
    const u64 k = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= nRecords) return;
    const u64 i = order[k];
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = tiles[t].records[i - tiles[t].firstRecord];
    if (!bamStored(r)) return;
    const u32 n = bamRecordBytes(tiles[t], r, o);
    if (offsets[k] + n <= capacity) bamWriteRecord(tiles[t], r, o, out + offsets[k]);
}
#endif

} // namespace isaac
