// FASTQ text -> BCL bytes on the device: io::FastqReader (lib/io/FastqReader.cpp:103-283, include/io/FastqReader.hh:144-210).
//
// The reference walks the text record by record.  Here the text is cut into lines in parallel (a line starts at a
// non-newline byte that follows a newline byte), and the one piece of sequential state the parser has -- which of the four
// roles (header, sequence, '+', qualities) a line plays, given that a sequence line starting with '+' is a zero-length read
// and takes the '+' role itself -- is a five-state automaton over the lines, evaluated with a scan over transition maps.
// Every header line then owns one record: one thread checks its lines as FastqReader::next does and converts the bases.
#pragma once
#include "types.h"

namespace isaac
{

enum { FQ_HEADER = 0, FQ_SEQUENCE = 1, FQ_PLUS = 2, FQ_QUALITY = 3, FQ_ERROR = 4 };
// transition maps packed 3 bits per source state
ISAAC_HD u32 fqMap(u32 h, u32 s, u32 p, u32 q, u32 e) { return h | (s << 3) | (p << 6) | (q << 9) | (e << 12); }
ISAAC_HD u32 fqApply(u32 map, u32 state) { return (map >> (3 * state)) & 7; }
// a line that starts with '+': closes a zero-length read when a sequence was expected; otherwise the '+' line / an ordinary
// header or quality line.  Any other line where '+' was expected is the reference's "+ sign not found where expected".
ISAAC_HD u32 fqLineMap(bool startsWithPlus)
{
    return startsWithPlus ? fqMap(FQ_SEQUENCE, FQ_HEADER, FQ_QUALITY, FQ_HEADER, FQ_ERROR)
                          : fqMap(FQ_SEQUENCE, FQ_PLUS, FQ_ERROR, FQ_HEADER, FQ_ERROR);
}
struct FqCompose   // scan operator: first a, then b
{
    ISAAC_HD u32 operator()(u32 a, u32 b) const
    {
        u32 r = 0;
        for (u32 s = 0; s < 5; ++s) r |= fqApply(b, fqApply(a, s)) << (3 * s);
        return r;
    }
};
static const u32 FQ_IDENTITY = 0 | (1 << 3) | (2 << 6) | (3 << 9) | (4 << 12);

ISAAC_HD bool fqIsNewLine(char c) { return '\n' == c || '\r' == c; }

// per record: what FastqReader::next + extractBcl make of it
enum { FQ_OK = 0, FQ_INCOMPLETE = 1, FQ_BAD_FORMAT = 2, FQ_BAD_LENGTH = 3 };
struct FqRecord { u64 recordEnd; u64 errorOffset; u32 status; u32 pad; };

#if defined(__HIPCC__)

__global__ void k_fq_line_starts(const char *text, u64 n, u8 *isStart)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    isStart[i] = (!fqIsNewLine(text[i]) && (0 == i || fqIsNewLine(text[i - 1]))) ? 1 : 0;
}
__global__ void k_fq_lines(const char *text, u64 n, const u64 *lineStart, u32 nLines, u64 *lineEnd, u32 *lineMap)
{
    const u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nLines) return;
    const u64 i = lineStart[l];
    u64 e = i;
    while (e < n && !fqIsNewLine(text[e])) ++e;     // lines are a few hundred bytes
    lineEnd[l] = e;
    lineMap[l] = fqLineMap('+' == text[i]);
}
__global__ void k_fq_headers(const u32 *stateMapBefore, u32 nLines, u32 *isHeader)
{
    const u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nLines) return;
    isHeader[l] = FQ_HEADER == fqApply(stateMapBefore[l], FQ_HEADER) ? 1u : 0u;
}

// oligo::getTranslator(true, INCORRECT_FASTQ_BASE) (include/oligo/Nucleotides.hh:41-59): 0..3, 4 = N, 5 = not a base
__device__ inline u32 fqTranslate(u8 c)
{
    switch (c)
    {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    case 'n': case 'N': return 4;
    default: return 5;
    }
}

// one thread per header line = per record
__global__ void k_fq_records(const char *text, u64 n, int final, int allowVariableLength, u32 readLength, const u64 *lineStart, const u64 *lineEnd, const u32 *lineMap,
                             const u32 *isHeader, const u32 *recordIndex, u32 nLines, u8 *bcl, u64 clusterStride, u32 maxClusters, FqRecord *records)
{
    const u32 h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nLines || !isHeader[h]) return;
    const u32 k = recordIndex[h];
    if (k >= maxClusters) return;
    FqRecord r; r.recordEnd = 0; r.errorOffset = 0; r.status = FQ_OK; r.pad = 0;
    const u32 plusMap = fqLineMap(true);
    // what FastqReader does when it runs out of buffer: fetch more (not final) or throw (final)
#define FQ_TRUNCATED(at) do { r.status = final ? u32(FQ_BAD_FORMAT) : u32(FQ_INCOMPLETE); r.errorOffset = (at); records[k] = r; return; } while (0)
    if (lineEnd[h] == n) FQ_TRUNCATED(n);                                  // findHeader: header line not terminated
    if (h + 1 >= nLines) FQ_TRUNCATED(n);                                  // findSequence: no sequence start
    u64 basesBegin = lineStart[h + 1], qualBegin, recordEnd;
    if (lineMap[h + 1] == plusMap)
    {   // zero-length read: the line is the '+' line, the "qualities" are empty and end where the next line starts
        if (lineEnd[h + 1] == n || h + 2 >= nLines) FQ_TRUNCATED(n);       // findQScores: no qscores start
        qualBegin = lineStart[h + 2]; recordEnd = qualBegin;
    }
    else
    {
        if (lineEnd[h + 1] == n) FQ_TRUNCATED(n);                          // sequence line not terminated
        if (h + 2 >= nLines) FQ_TRUNCATED(n);                              // no '+' line
        if (lineMap[h + 2] != plusMap) { r.status = FQ_BAD_FORMAT; r.errorOffset = lineStart[h + 2]; records[k] = r; return; }   // "+ sign not found where expected"
        if (lineEnd[h + 2] == n || h + 3 >= nLines) FQ_TRUNCATED(n);       // no qscores start
        qualBegin = lineStart[h + 3]; recordEnd = lineEnd[h + 3];
        if (recordEnd == n && !final) FQ_TRUNCATED(n);                     // the quality line may continue in the next piece
    }
#undef FQ_TRUNCATED
    r.recordEnd = recordEnd;
    // extractBcl (FastqReader.hh:144-210): walks the quality string; the base string is read alongside whatever its length
    u8 *out = bcl + u64(k) * clusterStride;
    u32 extracted = 0;
    for (u64 q = qualBegin, b = basesBegin; q != recordEnd && extracted < readLength; ++q, ++b, ++extracted)
    {
        const u32 base = fqTranslate(u8(text[b]));
        if (4 == base) out[extracted] = 0;
        else if (5 == base) { r.status = FQ_BAD_FORMAT; r.errorOffset = b; records[k] = r; return; }
        else
        {
            const u8 quality = u8(u8(text[q]) - 33);
            if (quality >= 64) { r.status = FQ_BAD_FORMAT; r.errorOffset = b; records[k] = r; return; }
            out[extracted] = u8(base | (quality << 2));
        }
    }
    if (extracted != readLength)
    {
        if (!allowVariableLength) { r.status = FQ_BAD_LENGTH; r.errorOffset = lineStart[h]; records[k] = r; return; }
        for (; extracted < readLength; ++extracted) out[extracted] = 0;
    }
    records[k] = r;
}

// the first record that is not simply converted
__global__ void k_fq_first_bad(const FqRecord *records, u32 nRecords, u32 *firstBad)
{
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nRecords && records[k].status != FQ_OK) atomicMin(firstBad, k);
}

#endif // __HIPCC__

} // namespace isaac
