// BAM alignment records on the device: what build::Build serialises per bin with bam::serializeAlignment over
// build::FragmentAccessorBamAdapter (include/bam/Bam.hh:257-345, include/build/FragmentAccessorBamAdapter.hh:62-377), in the order of
// PackedFragmentBuffer::orderForBam (include/build/PackedFragmentBuffer.hh:149-176), with duplicate marking and gap realignment when asked
// for (further down), and --bam-exclude-tags ZX,ZY (the default tag set: SM AS RG NM BC, OC on realigned records).
//
// The records of all tiles of a call are ordered together: 128-bit keys (position | global cluster id, unmapped, second read) sorted by
// two stable radix passes; record sizes are scanned in that order; every record is then written where it belongs, one thread each.
#pragma once
#include "cluster_ops.h"
#include "../../include/isaac_gpu.h"
#include "realign.h"

namespace isaac
{

// one tile's worth of isaac_gpu_select output plus what names its reads
struct BamTile
{
    const u8 *bcl; const FragmentRecord *records; const u32 *cigars; u64 firstRecord;   // index of the tile's first record among all records of the call
    u32 nRecords, nameLength; char name[64];                                             // "<flowcell>:<lane>:<tile>:" (FragmentAccessorBamAdapter::readName)
    const u32 *cigarsAlt;                                                                // CIGARs of realigned records (RECORD_CIGAR_REALIGNED), else NULL
    const FragmentRecord *recordsOriginal;                                               // with gap realignment: the records as the caller gave them (`records` is the stage's copy), else NULL
    DevTls tls;                                                                          // GapRealigner::updatePairDetails: the statistics of the tile's barcode
    char readGroup[28]; u32 readGroupLength;                                             // RG:Z of the tile's records: the barcode index of its lane (FragmentAccessorBamAdapter.hh:283-299)
};
struct BamOptions { u32 nReads, readLength[2], readOffset[2], clusterLength, forcedDodgyAlignmentScore, pessimisticMapQ, barcodeLength, readGroupLength; char barcode[64], readGroup[64];
                    u32 markDuplicates, keepDuplicates, realignGaps; RealignParams realign; DevTls tls;
                    u32 binFilter, binFirstContig, binEndContig, binUnaligned;        // isaac_bam_options::bin_*: which records the call writes
                    u64 binFirstPosition, binEndPosition;                             // binFilter 2: the bin is [first, end) in ReferencePosition values
                    isaac_bam_index_entry *indexEntries; };                            // isaac_bam_options::index_entries_dev

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

// does the record belong to the bin the call writes (isaac_bam_options::bin_*)?  The others are there as mates.
ISAAC_HD bool bamInBin(const FragmentRecord &r, const BamOptions &o)
{
    if (!o.binFilter) return true;
    if (bamUnalignedBin(r)) return 0 != o.binUnaligned;
    if (2 == o.binFilter) return r.fStrandPosition >= o.binFirstPosition && r.fStrandPosition < o.binEndPosition;     // ReferencePosition values order by contig, then position
    const u32 contig = refposContig(r.fStrandPosition);
    return contig >= o.binFirstContig && contig < o.binEndContig;
}
// The part of contig `contig` the call's bin covers, as BinSorter hands it to GapRealigner::realign (lib/build/BinSorter.cpp:405-417: binStartPos, binEndPos):
// the whole contig unless the bin is a range of positions that begins or ends inside it
ISAAC_HD void bamBinRange(const BamOptions &o, const DevReference &R, u32 contig, u64 &binStartPos, u64 &binEndPos)
{
    binStartPos = refpos(contig, 0); binEndPos = refpos(contig, contigLength(R, contig));
    if (2 == o.binFilter) { binStartPos = imax(binStartPos, o.binFirstPosition); binEndPos = imin(binEndPos, o.binEndPosition); }
}
// The BCL bytes of the cluster a tile's record `local` belongs to: records come in cluster order, n_reads per cluster, so the cluster's place in
// the tile's buffers is its records' -- which is its cluster id in the buffers of an isaac_gpu_select call, and stays right for the compacted
// tiles of isaac_gpu_bin_tile, where the id (the read's name) and the place differ.
ISAAC_HD const u8 *bamClusterBcl(const BamTile &t, u64 local, const BamOptions &o) { return t.bcl + (local / o.nReads) * o.clusterLength; }

ISAAC_HD u32 bamReadNameLength(const BamTile &t, const FragmentRecord &r) { return t.nameLength + decimalDigits(r.clusterId) + 2; }   // + ":0"
// Cigar::toString (include/alignment/Cigar.hh:74-95) of n operations: its length, and its character k (0 at and past the end)
ISAAC_HD u32 bamCigarStringLength(const u32 *c, u32 n) { u32 chars = 0; for (u32 k = 0; k < n; ++k) chars += decimalDigits(c[k] >> 4) + 1; return chars; }
ISAAC_HD u8 bamCigarStringChar(const u32 *c, u32 n, u32 at)
{
    for (u32 k = 0; k < n; ++k)
    {
        const u32 digits = decimalDigits(c[k] >> 4);
        if (at < digits) { u32 v = c[k] >> 4; for (u32 d = digits - 1 - at; d; --d) v /= 10; return u8('0' + v % 10); }
        if (at == digits) { const u32 code = c[k] & 15; return u8("MIDNSHP=X?"[code < 9 ? code : 9]); }
        at -= digits + 1;
    }
    return 0;
}
// original: the record before gap realignment (BamTile::recordsOriginal), or NULL
ISAAC_HD u32 bamRecordBytes(const BamTile &t, const FragmentRecord &r, const BamOptions &o, const FragmentRecord *original = nullptr)
{
    const bool aligned = !(r.flags & 2);
    u32 n = 4 + 32 + bamReadNameLength(t, r) + 1 + (aligned ? 4 * u32(r.cigarLength) : 0) + (r.readLength + 1) / 2 + r.readLength;
    if (DODGY_ALIGNMENT_SCORE != r.alignmentScore) n += 7;                                                    // SM:i
    if ((r.flags & 256) && DODGY_ALIGNMENT_SCORE != r.templateAlignmentScore) n += 7;                        // AS:i
    n += 3 + t.readGroupLength + 1;                                                                            // RG:Z
    n += 7;                                                                                                    // NM:i
    n += 3 + o.barcodeLength + 1;                                                                              // BC:Z
    if (original && (r.reserved & RECORD_CIGAR_REALIGNED)) n += 3 + bamCigarStringLength(t.cigars + original->cigarOffset, original->cigarLength) + 1;   // OC:Z
    return n;
}

// Where the parts of one record lie and the values that are the same for all of its bytes.  A record is written by a whole wave, lane j
// producing bytes j, j + 64, ...: every store instruction covers 64 consecutive bytes.
struct BamLayout
{
    u32 words[9];                                    // block_size, refID, pos, bin_mq_nl, flag_nc, l_seq, next_refID, next_pos, tlen
    u32 nameBegin, digitsBegin, nameTail, cigarBegin, seqBegin, qualBegin, tagBegin, total;
    u32 digits, clusterId, nCigar, readLength, reverse, smAt, asAt, rgAt, nmAt, bcAt, ocAt;   // tag offsets relative to tagBegin (~0u: absent)
    u32 sm, as, nm, nOriginalCigar;
    const char *readGroup; u32 readGroupLength;                                          // BamTile::readGroup
    const u8 *bcl; const u32 *cigar, *originalCigar;                                     // originalCigar: getFragmentOC (FragmentAccessorBamAdapter.hh:182-198), realigned fragments only
};

// bam::serializeAlignment (Bam.hh:257-345): the fixed part and the section boundaries
// local: the record's place in the tile
ISAAC_HD void bamLayout(const BamTile &t, const FragmentRecord &r, u64 local, const BamOptions &o, BamLayout &l, bool duplicate = false, const FragmentRecord *original = nullptr)
{
    const bool aligned = !(r.flags & 2), unalignedBin = bamUnalignedBin(r), paired = r.flags & 1;
    // FragmentAccessorBamAdapter::operator(): aligned fragments and shadows carry the bin index position, unaligned templates NoMatch
    const i32 refId = unalignedBin ? -1 : i32(refposContig(r.fStrandPosition)), pos = unalignedBin ? -1 : i32(refposPosition(r.fStrandPosition));
    const u32 nameLength = bamReadNameLength(t, r), observed = r.observedLength;
    l.nCigar = aligned ? r.cigarLength : 0;
    l.total = bamRecordBytes(t, r, o, original);
    const bool noMate = !paired || ((r.flags & 2) && (r.flags & 4));
    l.words[0] = l.total - 4; l.words[1] = u32(refId); l.words[2] = u32(pos);
    l.words[3] = (bamReg2bin(u32(pos), u32(pos) + (observed ? observed : 1)) << 16) | (bamMapq(r, o) << 8) | (nameLength + 1);
    l.words[4] = ((bamFlag(r) | (duplicate ? 1024u : 0u)) << 16) | (l.nCigar & 0xffff);      // bit 10: FragmentAccessorBamAdapter.hh:357
    l.words[5] = r.readLength;
    l.words[6] = noMate ? u32(-1) : u32(refposContig(r.mateFStrandPosition));
    l.words[7] = noMate ? u32(-1) : u32(refposPosition(r.mateFStrandPosition));
    l.words[8] = u32(r.bamTlen);
    l.digits = decimalDigits(r.clusterId); l.clusterId = r.clusterId; l.readLength = r.readLength; l.reverse = (r.flags & 8) ? 1 : 0;
    l.nameBegin = 36; l.digitsBegin = l.nameBegin + t.nameLength; l.nameTail = l.digitsBegin + l.digits; l.cigarBegin = l.nameTail + 3;
    l.seqBegin = l.cigarBegin + 4 * l.nCigar; l.qualBegin = l.seqBegin + (l.readLength + 1) / 2; l.tagBegin = l.qualBegin + l.readLength;
    u32 at = 0;
    const bool sm = DODGY_ALIGNMENT_SCORE != r.alignmentScore, as = (r.flags & 256) && DODGY_ALIGNMENT_SCORE != r.templateAlignmentScore;
    l.smAt = sm ? at : ~0u; at += sm ? 7 : 0;
    l.asAt = as ? at : ~0u; at += as ? 7 : 0;
    l.rgAt = at; at += 3 + t.readGroupLength + 1; l.readGroup = t.readGroup; l.readGroupLength = t.readGroupLength;
    l.nmAt = at; at += 7;
    l.bcAt = at; at += 3 + o.barcodeLength + 1;
    const bool oc = original && (r.reserved & RECORD_CIGAR_REALIGNED);
    l.ocAt = oc ? at : ~0u; l.originalCigar = oc ? t.cigars + original->cigarOffset : nullptr; l.nOriginalCigar = oc ? original->cigarLength : 0;
    l.sm = r.alignmentScore; l.as = r.templateAlignmentScore; l.nm = r.editDistance;
    const u32 readIndex = (r.flags & 64) && paired ? 1u : 0u;
    l.bcl = bamClusterBcl(t, local, o) + o.readOffset[readIndex];
    l.cigar = ((r.reserved & RECORD_CIGAR_REALIGNED) ? t.cigarsAlt : t.cigars) + r.cigarOffset;
}

// the base as FragmentCollector::storeBclAndCigar keeps it (reverse-complemented for reverse alignments, FragmentCollector.cpp:84-96) ...
ISAAC_HD u8 bamStoredBcl(const BamLayout &l, u32 i) { const u8 b = l.reverse ? l.bcl[l.readLength - 1 - i] : l.bcl[i]; return (b & 0xfc) ? (l.reverse ? u8((b & 0xfc) | (3 - (b & 3))) : b) : u8(0); }
// ... and as the adapter converts it (bamBaseFromBclByte :218-221)
ISAAC_HD u32 bamBase4(u8 stored) { return (stored & 0xfc) ? 1u << (stored & 3) : 15u; }
ISAAC_HD u8 bamIntTagByte(char a, char b, u32 v, u32 k) { return k == 0 ? u8(a) : k == 1 ? u8(b) : k == 2 ? u8('i') : u8(v >> (8 * (k - 3))); }
ISAAC_HD u8 bamStringTagByte(char a, char b, const char *s, u32 n, u32 k) { return k == 0 ? u8(a) : k == 1 ? u8(b) : k == 2 ? u8('Z') : k - 3 < n ? u8(s[k - 3]) : u8(0); }

// byte j of the record; `text`: the strings that go into it; `stored` (optional): the read's bases as bamStoredBcl gives them, staged by the caller; `cigar`: l.cigar or a staged copy
struct BamStrings { const char *namePrefix, *barcode; u32 barcodeLength; };   // BamTile::name, BamOptions::barcode or staged copies
ISAAC_HD u8 bamRecordByte(const BamStrings &text, const BamLayout &l, u32 j, const u8 *stored, const u32 *cigar)
{
    if (j < l.nameBegin)
    {
        const u32 w = j >> 2;
        u32 v = l.words[0];
        for (u32 k = 1; k < 9; ++k) v = (w == k) ? l.words[k] : v;
        return u8(v >> (8 * (j & 3)));
    }
    if (j < l.digitsBegin) return u8(text.namePrefix[j - l.nameBegin]);
    if (j < l.nameTail)
    {
        u32 v = l.clusterId;
        for (u32 k = l.nameTail - 1 - j; k; --k) v /= 10;
        return u8('0' + v % 10);
    }
    if (j < l.cigarBegin) return j == l.nameTail ? u8(':') : j == l.nameTail + 1 ? u8('0') : u8(0);
    if (j < l.seqBegin) { const u32 k = j - l.cigarBegin; return u8(cigar[k >> 2] >> (8 * (k & 3))); }
    if (j < l.qualBegin)
    {
        const u32 i = 2 * (j - l.seqBegin);
        const u32 hi = bamBase4(stored ? stored[i] : bamStoredBcl(l, i)), lo = i + 1 < l.readLength ? bamBase4(stored ? stored[i + 1] : bamStoredBcl(l, i + 1)) : 0u;
        return u8((hi << 4) | lo);
    }
    if (j < l.tagBegin) return u8((stored ? stored[j - l.qualBegin] : bamStoredBcl(l, j - l.qualBegin)) >> 2);               // bamQualFromBclByte :228-230
    const u32 k = j - l.tagBegin;
    if (k >= l.ocAt) return k - l.ocAt == 0 ? u8('O') : k - l.ocAt == 1 ? u8('C') : k - l.ocAt == 2 ? u8('Z') : bamCigarStringChar(l.originalCigar, l.nOriginalCigar, k - l.ocAt - 3);
    if (k >= l.bcAt) return bamStringTagByte('B', 'C', text.barcode, text.barcodeLength, k - l.bcAt);
    if (k >= l.nmAt) return bamIntTagByte('N', 'M', l.nm, k - l.nmAt);
    if (k >= l.rgAt) return bamStringTagByte('R', 'G', l.readGroup, l.readGroupLength, k - l.rgAt);
    if (l.asAt != ~0u && k >= l.asAt) return bamIntTagByte('A', 'S', l.as, k - l.asAt);
    return bamIntTagByte('S', 'M', l.sm, k);
}

#if defined(__HIPCC__)
// tile of global record index i: the last one that starts at or before it (the tiles of a bin are one per tile of the run: bisection)
__device__ inline u32 bamTileOf(const BamTile *tiles, u32 nTiles, u64 i)
{
    u32 lo = 0, hi = nTiles;
    while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if (tiles[mid].firstRecord <= i) lo = mid; else hi = mid; }
    return lo;
}

// orderForBam as a 128-bit key: hi = the bin index position (unaligned templates and dropped records last), lo = global cluster id,
// unmapped, second read
// duplicate: the verdicts of k_dup_mark, or NULL; a duplicate is left out like a record that was never stored when --keep-duplicates is off
__global__ void k_bam_keys(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, const u8 *duplicate, u64 *keyHi, u64 *keyLo, u32 *index, u32 *bytes)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = tiles[t].records[i - tiles[t].firstRecord];
    const bool stored = bamStored(r) && bamInBin(r, o) && !(duplicate && duplicate[i] && !o.keepDuplicates);
    keyHi[i] = !stored ? ~u64(0) : bamUnalignedBin(r) ? ~u64(0) - 1 : r.fStrandPosition;
    keyLo[i] = ((u64(r.tile) * INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE + r.clusterId) << 2) | ((r.flags & 2) ? 2u : 0u) | ((r.flags & 64) ? 1u : 0u);
    index[i] = u32(i);
    bytes[i] = stored ? bamRecordBytes(tiles[t], r, o, tiles[t].recordsOriginal ? tiles[t].recordsOriginal + (i - tiles[t].firstRecord) : nullptr) : 0;
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
// ---- duplicate marking (--mark-duplicates 1, --keep-duplicates): BinSorter::resolveDuplicates (lib/build/BinSorter.cpp:293-330) over all
// records of the call.  The ends of pairs that have a bin position are ranked by the reference's comparators (include/build/
// DuplicateFragmentIndexFiltering.hh:37-208: forward-strand ends by f-strand position, reverse-strand ends and shadows by their anchor; then
// the mate's anchor and orientation, the library, the template's rank descending, the global cluster id) and every end that equals the
// best one before it in position, mate anchor, mate orientation and library -- but belongs to another cluster -- is a duplicate
// (DuplicatePairEndFilter.hh:45-107).  One library (one barcode); bins as wide as a contig, so that mates with equal anchors share a
// storage bin.  Here: five key arrays, five stable radix passes, one pass over the sorted ends.
// io::FragmentIndexAnchor (include/io/Fragment.hh:490-506)
__device__ inline u64 dupAnchor(const FragmentRecord &h, const u8 *readBcl)
{
    if (!(h.flags & 2)) return (h.flags & 8) ? h.fStrandPosition + (u64(imax(h.observedLength, 1u) - 1) << 1) : h.fStrandPosition;
    u64 packed = 0;                                                        // oligo::pack32BclBases of a shadow
    for (u32 i = 0; i < 32 && i < h.readLength; ++i) packed |= u64(readBcl[i] & 3) << (2 * i);
    return packed;
}
// keySmall: 0 for records that take no part, else 4 | 8 (reverse-strand end or shadow) | mate_.info_ (shadow | reverse << 1)
__global__ void k_dup_keys(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, u64 *keyPrimary, u64 *keyMate, u64 *keyRank, u64 *keyCluster, u64 *keySmall, u32 *index)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const u64 local = i - tiles[t].firstRecord;
    const FragmentRecord &h = tiles[t].records[local];
    index[i] = u32(i);
    u64 primary = 0, mate = 0, rank = 0, cluster = 0, small = 0;
    // (a bin is filtered by itself, BinSorter::resolveDuplicates: the records that are in the call's tiles as mates of the bin's own are not part of it)
    if (bamStored(h) && (h.flags & 1) && !bamUnalignedBin(h) && bamInBin(h, o))
    {
        const FragmentRecord &m = tiles[t].records[local ^ 1];          // records come in cluster order, read 0 before read 1
        const u8 *clusterBcl = bamClusterBcl(tiles[t], local, o);
        const u32 readIndex = (h.flags & 64) ? 1u : 0u;
        // io::getTemplateDuplicateRank (Fragment.hh:66-71); an N has quality 2 (Read.cpp:56-69)
        u32 quality = 0;
        for (u32 b = 0; b < o.clusterLength; ++b) { const u8 v = clusterBcl[b]; quality += (v & 0xfc) ? u32(v >> 2) : 2u; }
        const u32 score = (h.reserved >> 16) == 0xffffu ? 0xffffffffu : (h.reserved >> 16);
        rank = (u64(quality) << 32) | u64(((u32(h.readLength) + m.readLength - (u32(h.editDistance) + m.editDistance)) << 16) | score);
        mate = dupAnchor(m, clusterBcl + o.readOffset[1 - readIndex]);
        const bool rs = (h.flags & 8) || (h.flags & 2);
        primary = rs ? dupAnchor(h, clusterBcl + o.readOffset[readIndex]) : h.fStrandPosition;
        cluster = u64(h.tile) * INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE + h.clusterId;
        small = 4u | (rs ? 8u : 0u) | ((h.flags & 4) ? 1u : 0u) | ((h.flags & 16) ? 2u : 0u);
    }
    keyPrimary[i] = primary; keyMate[i] = mate; keyRank[i] = ~rank /* higher rank first */; keyCluster[i] = cluster; keySmall[i] = small;
}
__global__ void k_dup_gather(const u64 *key, const u32 *order, u64 n, u64 *out) { const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; if (i < n) out[i] = key[order[i]]; }
// order: the ends sorted by (small, primary, mate anchor | rank descending, cluster).  The first end of a group is the best; the walk is
// DuplicatePairEndFilter's loop: an end of another cluster than the last survivor is a duplicate, an end of the same cluster survives
// and takes over ("both ends of a pair facing the same way at the same position").
__global__ void k_dup_mark(const u32 *order, u64 n, const u64 *keyPrimary, const u64 *keyMate, const u64 *keyCluster, const u64 *keySmall, u8 *duplicate)
{
    const u64 k = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const u32 i = order[k];
    const u64 small = keySmall[i];           // (the verdicts are zeroed before the launch: a group's head writes the flags of its followers)
    if (!small) return;
    const u64 primary = keyPrimary[i], mate = keyMate[i];
    if (k) { const u32 p = order[k - 1]; if (keySmall[p] == small && keyPrimary[p] == primary && keyMate[p] == mate) return; }     // not the head of its group
    u64 last = keyCluster[i];
    for (u64 j = k + 1; j < n; ++j)
    {
        const u32 e = order[j];
        if (keySmall[e] != small || keyPrimary[e] != primary || keyMate[e] != mate) break;
        const u64 c = keyCluster[e];
        if (c != last) duplicate[e] = 1; else last = c;
    }
}

// ---- gap realignment (--realign-gaps sample): BinSorter::collectGaps / realignGaps (lib/build/BinSorter.cpp:387-417) over the call's records,
// every contig one bin.  The records are the stage's private copy (the caller's stay as they are).
//   k_realign_count / k_realign_collect   the gaps of every stored aligned fragment (duplicates included, as the reference walks the bin's data)
//   host                                  RealignerGaps::finalizeGaps with std::sort of the same libstdc++, contig by contig (ties among deletions
//                                         that end at the same place keep the order the reference's sort gives them)
//   k_realign                             GapRealigner::realign per kept fragment (realign.h)
//   k_realign_pairs                       GapRealigner::updatePairDetails once both ends are final
__device__ inline const u32 *bamRecordCigar(const BamTile &t, const FragmentRecord &r) { return ((r.reserved & RECORD_CIGAR_REALIGNED) ? t.cigarsAlt : t.cigars) + r.cigarOffset; }
__device__ inline bool realignHasGaps(const FragmentRecord &r, const BamOptions &o) { return bamStored(r) && !(r.flags & 2) && !bamUnalignedBin(r) && r.gapCount && bamInBin(r, o); }
__global__ void k_realign_count(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, u32 *counts)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = tiles[t].records[i - tiles[t].firstRecord];
    u32 n = 0;
    if (realignHasGaps(r, o)) { const u32 *c = bamRecordCigar(tiles[t], r); for (u32 k = 0; k < r.cigarLength; ++k) { const u32 code = cigarCode(c[k]); n += (OP_INSERT == code || OP_DELETE == code); } }
    counts[i] = n;
}
// RealignerGaps::addGaps (GapRealigner.hh:54-104)
__global__ void k_realign_collect(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, const u32 *offsets, RealignGap *gaps)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = tiles[t].records[i - tiles[t].firstRecord];
    if (!realignHasGaps(r, o)) return;
    const u32 *c = bamRecordCigar(tiles[t], r);
    u64 pos = r.fStrandPosition; u32 at = offsets[i];
    for (u32 k = 0; k < r.cigarLength; ++k)
    {
        const u32 length = cigarLen(c[k]), code = cigarCode(c[k]);
        if (OP_ALIGN == code) pos = rpPlus(pos, length);
        else if (OP_INSERT == code) { RealignGap g; g.pos = pos; g.length = -i32(length); g.pad = 0; gaps[at++] = g; }
        else if (OP_DELETE == code) { RealignGap g; g.pos = pos; g.length = i32(length); g.pad = 0; gaps[at++] = g; pos = rpPlus(pos, length); }
    }
}
// The fragments the realigner has to look at: those of the bin that pass its cheap tests and have a gap of the bin's list inside their range (GapRealigner::findGaps
// finding none is how the reference leaves nearly every fragment alone).  A thread per record with a few registers; k_realign -- a thread per fragment with
// the realigner's whole state, ten gaps and three CIGARs of scratch -- then runs over the list only (round 5: it ran over every record, 19 ms for the
// 3.4 M records of a bin of which a few thousand have anything to try).  changed[] is cleared for every record.
// (wanted[i] = 1 for the listed records: the list is made from these flags by a scan, in record order -- appended with an atomic counter it came out in the
// order the waves happened to run, and k_realign's reads of neighbouring records no longer shared lines: at thirty-fold depth, where a third of a bin's
// fragments have a gap in reach, the stage was slower than without the filter)
__global__ void k_realign_filter(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, RealignerGapsView gapsView, const u8 *duplicate, const FragmentRecord *records /* the copy */,
                                 u32 *wanted, u8 *changed)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    changed[i] = 0;
    wanted[i] = 0;
    const u32 t = bamTileOf(tiles, nTiles, i);
    const FragmentRecord &r = records[i];
    if (!bamStored(r) || (r.flags & 2) || bamUnalignedBin(r) || !r.editDistance || !bamInBin(r, o)) return;
    if (duplicate && duplicate[i] && !o.keepDuplicates) return;
    const u32 *cigar = bamRecordCigar(tiles[t], r);
    RealignIndex index = { r.fStrandPosition, cigar, cigar + r.cigarLength };
    const RealignBounds bounds = rgBounds(index);
    if (rgAnyGap(gapsView, bounds.beginPos, bounds.endPos)) wanted[i] = 1;
}
__global__ void k_realign_list(const u32 *wanted, const u32 *offsets, u64 nRecords, u32 *list)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < nRecords && wanted[i]) list[offsets[i]] = u32(i);
}
// changed[i]: the fragment was realigned; its new CIGAR took words from the pool's bump counter
__global__ void k_realign(BamTile *tiles, u32 nTiles, const u32 *list, u32 listCount, BamOptions o, DevReference R, RealignerGapsView gapsView, const u8 *duplicate, FragmentRecord *records /* the copy */,
                          u32 *pool, u32 poolCap, u32 *poolNext, u8 *changed)
{
    const u64 entry = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (entry >= listCount) return;
    const u64 i = list[entry];
    changed[i] = 0;
    const u32 t = bamTileOf(tiles, nTiles, i);
    FragmentRecord &r = records[i];
    if (!bamStored(r) || (r.flags & 2) || bamUnalignedBin(r) || !r.editDistance || !bamInBin(r, o)) return;
    if (duplicate && duplicate[i] && !o.keepDuplicates) return;                 // not in the bin's index any more
    RealignCtx x; x.R = &R; x.P = o.realign;
    RealignFragment f;
    f.fStrandPosition = r.fStrandPosition; f.mateFStrandPosition = r.mateFStrandPosition; f.observedLength = r.observedLength; f.flags = r.flags;
    f.lowClipped = r.lowClipped; f.highClipped = r.highClipped; f.alignmentScore = r.alignmentScore; f.templateAlignmentScore = r.templateAlignmentScore;
    f.readLength = r.readLength; f.editDistance = r.editDistance;
    const u32 readIndex = ((r.flags & 64) && (r.flags & 1)) ? 1u : 0u;
    f.bcl = bamClusterBcl(tiles[t], i - tiles[t].firstRecord, o) + o.readOffset[readIndex];
    const u32 *cigar = bamRecordCigar(tiles[t], r);
    RealignIndex index = { r.fStrandPosition, cigar, cigar + r.cigarLength };
    const u32 contig = refposContig(r.fStrandPosition);
    RealignCigar result;
    u64 binStartPos, binEndPos;
    bamBinRange(o, R, contig, binStartPos, binEndPos);
    if (!realignFragment(x, gapsView, binStartPos, binEndPos, index, f, result)) return;
    const u32 at = atomicAdd(poolNext, result.n);
    if (at + result.n > poolCap) return;                                         // no room: the fragment keeps its alignment (sized so that this does not happen)
    for (u32 k = 0; k < result.n; ++k) pool[at + k] = result.words[k];
    r.fStrandPosition = f.fStrandPosition; r.observedLength = f.observedLength; r.editDistance = f.editDistance; r.cigarLength = u16(result.n); r.cigarOffset = at;
    r.reserved |= RECORD_CIGAR_REALIGNED;
    changed[i] = 1;
}
// GapRealigner::updatePairDetails (:267-318).  The reference brings a pair up to date whenever one of its ends has been realigned, the ends of a
// bin in the order of its index: reverse-strand ends and shadows first, then forward-strand ends, each list in the duplicate filter's order.
// With both ends final that is: TLEN and mate positions from the final alignments, proper-pair from checkModel(end realigned last, its mate).
__global__ void k_realign_pairs(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, FragmentRecord *records, const u8 *changed)
{
    const u64 unit = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (1 == o.nReads)
    {   // single-ended: !index.hasMate() (:272-276)
        if (unit < nRecords && changed[unit]) { FragmentRecord &r = records[unit]; r.bamTlen = r.bamTlen < 0 ? -i32(r.observedLength) + 1 : i32(r.observedLength) - 1; }
        return;
    }
    const u64 i = 2 * unit;                       // a pair: records 2k (read 1) and 2k + 1 (read 2) of a tile
    if (i + 1 >= nRecords) return;
    FragmentRecord &a = records[i];
    FragmentRecord &b = records[i + 1];
    if (!changed[i] && !changed[i + 1]) return;
    const u32 t = bamTileOf(tiles, nTiles, i);
    // which end was realigned last
    bool aLast = changed[i] && !changed[i + 1];
    if (changed[i] && changed[i + 1])
    {
        const bool aForward = !(a.flags & 8), bForward = !(b.flags & 8);
        if (aForward != bForward) aLast = aForward;
        else
        {
            const u8 *clusterBcl = bamClusterBcl(tiles[t], i - tiles[t].firstRecord, o);
            const u64 pa = aForward ? a.fStrandPosition : dupAnchor(a, clusterBcl), pb = bForward ? b.fStrandPosition : dupAnchor(b, clusterBcl + o.readOffset[1]);
            aLast = pa > pb;      // later in the list's order; equal keys: the second read
        }
    }
    FragmentRecord &f = aLast ? a : b, &m = aLast ? b : a;
    const u64 fragmentBeginPos = f.fStrandPosition, fragmentEndPos = rpPlus(fragmentBeginPos, f.observedLength), mateBeginPos = m.fStrandPosition, mateEndPos = rpPlus(mateBeginPos, m.observedLength);
    f.mateFStrandPosition = m.fStrandPosition;
    f.bamTlen = rgTlen(fragmentBeginPos, fragmentEndPos, mateBeginPos, mateEndPos, 0 != (f.flags & 32));
    m.bamTlen = -f.bamTlen;
    m.mateFStrandPosition = f.fStrandPosition;
    Cand cf, cm; candInit(cf, 0); candInit(cm, 1);
    cf.contigId = refposContig(f.fStrandPosition); cf.position = i64(refposPosition(f.fStrandPosition)); cf.reverse = (f.flags & 8) ? 1 : 0; cf.observedLength = f.observedLength; cf.cigarLength = 1;
    cm.contigId = refposContig(m.fStrandPosition); cm.position = i64(refposPosition(m.fStrandPosition)); cm.reverse = (m.flags & 8) ? 1 : 0; cm.observedLength = m.observedLength; cm.cigarLength = 1;
    const bool proper = TLS_NOMINAL == tlsCheckModel(tiles[t].tls, cf, cm);
    f.flags = (f.flags & ~256u) | (proper ? 256u : 0u); m.flags = (m.flags & ~256u) | (proper ? 256u : 0u);
}

// One workgroup per BAM_CHUNK_RECORDS consecutive records of the file.  Everything is done by threads that each own a small piece of a
// record, so that a vector instruction works on 64 pieces at once (the first versions gave a record to a whole wave, which then issued the
// few hundred instructions of its irregular parts -- name, tags -- for that one record: 1.4 ns per record, instruction issue bound):
//   A  a thread per record: order entry -> record -> layout (LDS)
//   B  a thread per piece: the fixed part | the read name | CIGAR + SM + AS | RG + NM + BC | every 16 bases (sequence nibbles and
//      qualities, straight from the BCL bytes, reverse-complemented for reverse alignments), written into the chunk's image in LDS
//   C  the image, a contiguous piece of the file, leaves with 16-byte stores on 16-byte boundaries
// Chunks whose records do not fit the image (CIGARs far beyond the usual) are written byte by byte with bamRecordByte.
#ifndef ISAAC_BAM_CHUNK_RECORDS
#define ISAAC_BAM_CHUNK_RECORDS 32
#endif
static const u32 BAM_STAGE_TILES = 32, BAM_CHUNK_RECORDS = ISAAC_BAM_CHUNK_RECORDS, BAM_SMALL_PIECES = 4;
struct BamChunkLds { u32 chunkBytes, segments; };     // dynamic LDS: image[chunkBytes + 16]; segments: 16-base pieces of the longest read
ISAAC_HD u32 bamChunkImageBytes(u32 maxReadLength, u32 nameBytes, u32 tagBytes) { return BAM_CHUNK_RECORDS * (36 + nameBytes + 4 * 40 + (maxReadLength + 1) / 2 + maxReadLength + tagBytes); }

__device__ inline void bamPut32(u8 *to, u32 v) { to[0] = u8(v); to[1] = u8(v >> 8); to[2] = u8(v >> 16); to[3] = u8(v >> 24); }
__device__ inline void bamPutIntTag(u8 *to, char a, char b, u32 v) { to[0] = u8(a); to[1] = u8(b); to[2] = u8('i'); bamPut32(to + 3, v); }
__device__ inline void bamPutStringTag(u8 *to, char a, char b, const char *s, u32 n) { to[0] = u8(a); to[1] = u8(b); to[2] = u8('Z'); for (u32 i = 0; i < n; ++i) to[3 + i] = u8(s[i]); to[3 + n] = 0; }

// one of the small pieces of a record (the same bytes bamRecordByte gives for those positions)
__device__ inline void bamWriteSmallPiece(u32 piece, const BamLayout &l, const char *name, const char *barcode, u32 barcodeLength, u8 *to)
{
    if (0 == piece) { for (u32 w = 0; w < 9; ++w) bamPut32(to + 4 * w, l.words[w]); }
    else if (1 == piece)
    {
        const u32 nameBegin = l.nameBegin, digitsBegin = l.digitsBegin, nameTail = l.nameTail;
        for (u32 i = nameBegin; i < digitsBegin; ++i) to[i] = u8(name[i - nameBegin]);
        u32 v = l.clusterId;
        for (u32 i = nameTail; i > digitsBegin; --i) { to[i - 1] = u8('0' + v % 10); v /= 10; }
        to[nameTail] = u8(':'); to[nameTail + 1] = u8('0'); to[nameTail + 2] = 0;
    }
    else if (2 == piece)
    {
        const u32 n = l.nCigar, cigarBegin = l.cigarBegin; const u32 *cigar = l.cigar;
        for (u32 i = 0; i < n; ++i) bamPut32(to + cigarBegin + 4 * i, cigar[i]);
        const u32 tagBegin = l.tagBegin, smAt = l.smAt, asAt = l.asAt;
        if (~0u != smAt) bamPutIntTag(to + tagBegin + smAt, 'S', 'M', l.sm);
        if (~0u != asAt) bamPutIntTag(to + tagBegin + asAt, 'A', 'S', l.as);
    }
    else
    {
        const u32 tagBegin = l.tagBegin;
        bamPutStringTag(to + tagBegin + l.rgAt, 'R', 'G', l.readGroup, l.readGroupLength);
        bamPutIntTag(to + tagBegin + l.nmAt, 'N', 'M', l.nm);
        bamPutStringTag(to + tagBegin + l.bcAt, 'B', 'C', barcode, barcodeLength);
        if (~0u != l.ocAt)
        {
            u8 *oc = to + tagBegin + l.ocAt; const u32 chars = l.total - (tagBegin + l.ocAt) - 4;
            oc[0] = u8('O'); oc[1] = u8('C'); oc[2] = u8('Z');
            for (u32 k = 0; k <= chars; ++k) oc[3 + k] = bamCigarStringChar(l.originalCigar, l.nOriginalCigar, k);
        }
    }
}

// stored positions [16 * segment, 16 * segment + 16) of the read: their qualities and the eight sequence bytes they make
__device__ inline void bamWriteBases(u32 segment, const BamLayout &l, u8 *to)
{
    const u32 L = l.readLength, s0 = 16 * segment;
    if (s0 >= L) return;
    const bool reverse = 0 != l.reverse;
    const u8 *bcl = l.bcl;
    u8 v[16];
#pragma unroll
    for (u32 b = 0; b < 16; ++b) v[b] = s0 + b < L ? (reverse ? bcl[L - 1 - s0 - b] : bcl[s0 + b]) : u8(0);
    u8 *qual = to + l.qualBegin + s0, *seq = to + l.seqBegin + s0 / 2;
#pragma unroll
    for (u32 b = 0; b < 16; b += 2)
    {
        // FragmentCollector::storeBclAndCigar (reverse complement for reverse alignments), then bamBaseFromBclByte / bamQualFromBclByte
        const u8 c0 = v[b], c1 = v[b + 1];
        const u8 st0 = (c0 & 0xfc) ? (reverse ? u8((c0 & 0xfc) | (3 - (c0 & 3))) : c0) : u8(0);
        const u8 st1 = (c1 & 0xfc) ? (reverse ? u8((c1 & 0xfc) | (3 - (c1 & 3))) : c1) : u8(0);
        if (s0 + b < L) { qual[b] = u8(st0 >> 2); seq[b / 2] = u8((bamBase4(st0) << 4) | (s0 + b + 1 < L ? bamBase4(st1) : 0u)); }
        if (s0 + b + 1 < L) qual[b + 1] = u8(st1 >> 2);
    }
}

__global__ void __launch_bounds__(256) k_bam_encode(const BamTile *tiles, u32 nTiles, u64 nRecords, BamOptions o, const u32 *order, const u64 *offsets, const u64 *bytes,
                                                    const u8 *duplicate, u8 *out, u64 capacity, BamChunkLds lds)
{
    extern __shared__ __attribute__((aligned(16))) u8 image[];
    __shared__ BamLayout layouts[BAM_CHUNK_RECORDS];
    __shared__ u32 tileOfRecord[BAM_CHUNK_RECORDS];            // ~0u: no bytes for this record
    __shared__ u64 tileFirst[BAM_STAGE_TILES];
    __shared__ u32 imageAt[BAM_CHUNK_RECORDS];                 // where the record starts in the chunk
    __shared__ char optionText[2][64];
    __shared__ u32 tileNames[BAM_STAGE_TILES][16];             // BamTile::name of every tile (when there are few enough)
    const bool tilesStaged = nTiles <= BAM_STAGE_TILES;
    if (tilesStaged) for (u32 t = threadIdx.x; t < nTiles; t += blockDim.x) tileFirst[t] = tiles[t].firstRecord;
    if (tilesStaged) for (u32 x = threadIdx.x; x < nTiles * 16; x += blockDim.x) tileNames[x / 16][x % 16] = reinterpret_cast<const u32 *>(tiles[x / 16].name)[x % 16];
    if (threadIdx.x < 64) optionText[0][threadIdx.x] = o.readGroup[threadIdx.x]; else if (threadIdx.x < 128) optionText[1][threadIdx.x - 64] = o.barcode[threadIdx.x - 64];
    const u64 k0 = u64(blockIdx.x) * BAM_CHUNK_RECORDS, k1 = k0 + BAM_CHUNK_RECORDS < nRecords ? k0 + BAM_CHUNK_RECORDS : nRecords;
    const u32 count = u32(k1 - k0);
    // the chunk's piece of the file: [begin, end); records that are left out have no bytes and sort last
    const u64 begin = offsets[k0], end = offsets[k1 - 1] + bytes[k1 - 1];
    const u32 shift = u32((reinterpret_cast<u64>(out) + begin) & 15);        // image[shift + x] mirrors file byte begin + x: equal alignment mod 16
    const bool viaLds = end - begin <= lds.chunkBytes;
    __syncthreads();
    // ---- A
    if (threadIdx.x < count)
    {
        const u64 i = order[k0 + threadIdx.x], at = offsets[k0 + threadIdx.x];
        u32 t;
        if (tilesStaged) { u32 lo = 0, hi = nTiles; while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tileFirst[mid] <= i) lo = mid; else hi = mid; } t = lo; }
        else t = bamTileOf(tiles, nTiles, i);
        const FragmentRecord r = tiles[t].records[i - tiles[t].firstRecord];
        BamLayout l;
        const bool dup = duplicate && duplicate[i];
        bool write = bamStored(r) && bamInBin(r, o) && !(dup && !o.keepDuplicates);
        if (write) { bamLayout(tiles[t], r, i - tiles[t].firstRecord, o, l, dup && o.markDuplicates, tiles[t].recordsOriginal ? tiles[t].recordsOriginal + (i - tiles[t].firstRecord) : nullptr); write = at + l.total <= capacity; }
        if (write) { layouts[threadIdx.x] = l; imageAt[threadIdx.x] = u32(at - begin); }
        if (write && o.indexEntries)
        {
            isaac_bam_index_entry e;
            e.offset = at; e.bytes = l.total; e.ref_id = i32(l.words[1]); e.pos = i32(l.words[2]); e.flag = l.words[4] >> 16; e.seq_length = l.words[5]; e.observed = (r.flags & 2) ? 0u : r.observedLength;
            o.indexEntries[k0 + threadIdx.x] = e;
        }
        tileOfRecord[threadIdx.x] = write ? t : ~0u;
    }
    __syncthreads();
    if (!viaLds)
    {   // the general form: a wave per record at a time, byte j of the record by lane j, j + 64, ...
        const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (u32 rec = w; rec < count; rec += 4)
        {
            if (~0u == tileOfRecord[rec]) continue;
            const BamLayout &l = layouts[rec];
            BamStrings text = { tiles[tileOfRecord[rec]].name, optionText[1], o.barcodeLength };
            u8 *to = out + begin + imageAt[rec];
            for (u32 j = lane; j < l.total; j += 64) to[j] = bamRecordByte(text, l, j, nullptr, l.cigar);
        }
        return;
    }
    // ---- B
    const u32 pieces = BAM_SMALL_PIECES + lds.segments;
    for (u32 x = threadIdx.x; x < count * pieces; x += blockDim.x)
    {
        const u32 rec = x / pieces, piece = x % pieces;
        const u32 t = tileOfRecord[rec];
        if (~0u == t) continue;
        u8 *to = image + shift + imageAt[rec];
        if (piece < BAM_SMALL_PIECES)
            bamWriteSmallPiece(piece, layouts[rec], tilesStaged ? reinterpret_cast<const char *>(tileNames[t]) : tiles[t].name, optionText[1], o.barcodeLength, to);
        else bamWriteBases(piece - BAM_SMALL_PIECES, layouts[rec], to);
    }
    __syncthreads();
    // ---- C: the part of [begin, end) that fits the caller's buffer: records are whole or absent, and the ones that did not fit lie at the end
    u64 stop = end <= capacity ? end : begin;
    if (end > capacity) for (u64 k = k0; k < k1; ++k) { const u64 e = offsets[k] + bytes[k]; if (e <= capacity) stop = e; }
    const u32 n = u32(stop - begin);
    u8 *g = out + begin;
    const u32 head = n < ((16 - shift) & 15) ? n : ((16 - shift) & 15);      // bytes before the first 16-byte boundary
    if (threadIdx.x < head) g[threadIdx.x] = image[shift + threadIdx.x];
    const u32 body = (n - head) / 16;
    const uint4 *from = reinterpret_cast<const uint4 *>(image + shift + head);
    uint4 *gto = reinterpret_cast<uint4 *>(g + head);
    for (u32 x = threadIdx.x; x < body; x += blockDim.x) gto[x] = from[x];
    const u32 tail = n - head - 16 * body;
    if (threadIdx.x < tail) g[head + 16 * body + threadIdx.x] = image[shift + head + 16 * body + threadIdx.x];
}
#endif


// ---- isaac_gpu_bin_tile: the records of a select call cut into one compact tile per bin (BinningFragmentStorage's job) -----------------------
// Bins are numbered in file order: every contig has a first bin (several small contigs may share one) and a contig too large for one bin goes on
// into the bins behind it at the cut positions the caller gives; the last bin takes the templates without a position.
static const u32 BIN_MAX = 65535;        // bins of a call; entries of records without a bin carry the key n_bins and sort behind all others
struct BinMap { const u32 *binOfContig; u32 nContigs; const u64 *cuts /* ascending ReferencePosition values: a further bin of the cut's contig starts there */; u32 nCuts; u32 nBins; };
// per bin b: firstEntry[b] (nBins + 1 entries), bclAt / recordsAt / cigarsAt (nBins each): one array of 4 nBins + 1 words in device memory
struct BinLayout { const u64 *firstEntry, *bclAt, *recordsAt, *cigarsAt; };
#if defined(__HIPCC__)
__device__ inline u32 binOfRecord(const FragmentRecord &r, const BinMap &m)
{
    if (!bamStored(r)) return m.nBins;
    if (bamUnalignedBin(r)) return m.nBins - 1;
    const u32 contig = refposContig(r.fStrandPosition);
    if (contig >= m.nContigs) return m.nBins;
    u32 bin = m.binOfContig[contig];
    if (m.nCuts)
    {   // the cuts of this contig at or before the record: those in [first cut of the contig, first cut beyond the position)
        const u64 contigStart = refpos(contig, 0), key = r.fStrandPosition & ~u64(1);
        u32 lo = 0, hi = m.nCuts;
        while (lo < hi) { const u32 mid = (lo + hi) / 2; if (m.cuts[mid] <= key) lo = mid + 1; else hi = mid; }
        const u32 upTo = lo;
        lo = 0; hi = upTo;
        while (lo < hi) { const u32 mid = (lo + hi) / 2; if (m.cuts[mid] < contigStart) lo = mid + 1; else hi = mid; }
        bin += upTo - lo;
    }
    return bin;
}
// entries 2c, 2c + 1: the bins cluster c has a stored record in (the second only when it differs from the first), else n_bins
__global__ void k_bin_entries(const FragmentRecord *records, u32 nClusters, u32 nReads, BinMap map, u16 *keys, u32 *values)
{
    const u32 cl = blockIdx.x * blockDim.x + threadIdx.x;
    if (cl >= nClusters) return;
    const u32 b0 = binOfRecord(records[u64(cl) * nReads], map);
    u32 b1 = 2 == nReads ? binOfRecord(records[u64(cl) * 2 + 1], map) : map.nBins;
    if (b1 == b0) b1 = map.nBins;
    // the smaller bin first, so that the entries without a bin sort behind everything
    keys[2 * u64(cl)] = u16(b0 < b1 ? b0 : b1); keys[2 * u64(cl) + 1] = u16(b0 < b1 ? b1 : b0);
    values[2 * u64(cl)] = cl; values[2 * u64(cl) + 1] = cl;
}
// words[k]: CIGAR words of sorted entry k's cluster (0 for entries without a bin; one more zero at the end for the scan); counts[b], counts[nBins + b]: entries and words of bin b.
// Sorted entries: a workgroup's entries are of a few neighbouring bins, one atomic per lane run of equal bins
__global__ void k_bin_words(const FragmentRecord *records, u32 nReads, const u16 *sortedKeys, const u32 *sortedValues, u64 nEntries, u32 nBins, u64 *words, unsigned long long *counts)
{
    const u64 k = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    u32 bin = nBins; u64 w = 0;
    if (k < nEntries)
    {
        bin = sortedKeys[k];
        if (bin < nBins)
        {
            const u32 cl = sortedValues[k];
            for (u32 r = 0; r < nReads; ++r) w += records[u64(cl) * nReads + r].cigarLength;
        }
        words[k] = w;
        if (k + 1 == nEntries) words[nEntries] = 0;
    }
    // the entries are sorted by bin: a wavefront's lanes hold runs of equal bins; the last lane of a run adds the run's totals
    const u32 lane = threadIdx.x & 63;
    u64 entries = bin < nBins ? 1 : 0, sum = w;
    for (u32 o = 1; o < 64; o <<= 1)
    {
        const u32 otherBin = __shfl_up(bin, o, 64); const u64 otherEntries = __shfl_up(entries, o, 64), otherSum = __shfl_up(sum, o, 64);
        if (lane >= o && otherBin == bin) { entries += otherEntries; sum += otherSum; }
    }
    const u32 nextBin = __shfl_down(bin, 1, 64);
    if (bin < nBins && (63 == lane || nextBin != bin)) { atomicAdd(&counts[bin], (unsigned long long)entries); atomicAdd(&counts[nBins + bin], (unsigned long long)sum); }
}
// the layout of the output: bin after bin, every array on a multiple of 64 bytes (one thread: a few hundred bins at most in practice)
__global__ void k_bin_layout(const u64 *counts, u32 nBins, u32 nReads, u32 clusterLength, u64 *layout /* firstEntry[nBins + 1] | bclAt | recordsAt | cigarsAt */, u64 *total)
{
    if (blockIdx.x || threadIdx.x) return;
    u64 at = 0, firstEntry = 0;
    u64 *first = layout, *bclAt = layout + nBins + 1, *recordsAt = bclAt + nBins, *cigarsAt = recordsAt + nBins;
    for (u32 b = 0; b < nBins; ++b)
    {
        const u64 m = counts[b], w = counts[nBins + b];
        first[b] = firstEntry; bclAt[b] = at;
        at = (at + m * clusterLength + 63) & ~u64(63); recordsAt[b] = at;
        at = (at + m * nReads * sizeof(FragmentRecord) + 63) & ~u64(63); cigarsAt[b] = at;
        at = (at + w * 4 + 63) & ~u64(63);
        firstEntry += m;
    }
    first[nBins] = firstEntry;
    total[0] = at; total[1] = firstEntry;
}
// 64 threads per entry: the cluster's BCL bytes, its records and their CIGAR words to the entry's place in its bin's part
__global__ void __launch_bounds__(256) k_bin_gather(const u8 *bcl, const FragmentRecord *records, const u32 *cigars, u32 nReads, u32 clusterLength, const u16 *sortedKeys, const u32 *sortedValues,
                                                    u64 nEntries, const u64 *wordsBefore, BinLayout layout, u8 *out)
{
    const u64 k = u64(blockIdx.x) * 4 + threadIdx.x / 64;
    const u32 lane = threadIdx.x & 63;
    if (k >= nEntries) return;
    const u32 bin = sortedKeys[k], cl = sortedValues[k];
    const u64 slot = k - layout.firstEntry[bin], wordAt = wordsBefore[k] - wordsBefore[layout.firstEntry[bin]];
    const u8 *src = bcl + u64(cl) * clusterLength; u8 *dst = out + layout.bclAt[bin] + slot * clusterLength;
    if (0 == (clusterLength & 3)) { for (u32 i = 4 * lane; i < clusterLength; i += 256) *reinterpret_cast<u32 *>(dst + i) = *reinterpret_cast<const u32 *>(src + i); }
    else for (u32 i = lane; i < clusterLength; i += 64) dst[i] = src[i];
    FragmentRecord *recordsOut = reinterpret_cast<FragmentRecord *>(out + layout.recordsAt[bin]) + slot * nReads;
    u32 *cigarsOut = reinterpret_cast<u32 *>(out + layout.cigarsAt[bin]);
    u64 w = wordAt;
    for (u32 r = 0; r < nReads; ++r)
    {
        const FragmentRecord &in = records[u64(cl) * nReads + r];
        const u32 n = in.cigarLength, from = in.cigarOffset;
        if (lane < 16) reinterpret_cast<u32 *>(recordsOut + r)[lane] = (lane == offsetof(FragmentRecord, cigarOffset) / 4) ? u32(w) : reinterpret_cast<const u32 *>(&in)[lane];
        for (u32 i = lane; i < n; i += 64) cigarsOut[w + i] = cigars[from + i];
        w += n;
    }
}
#endif

} // namespace isaac
