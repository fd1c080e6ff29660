// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
//
// FASTQ text -> BCL bytes of a tile: io::FastqReader (lib/io/FastqReader.cpp:103-330, include/io/FastqReader.hh:144-210) and
// io::FastqLoader::loadSingleRead (include/io/FastqLoader.hh) over a memory buffer instead of a (gzip) stream.
// Parity unpinned: the reference has no unit test for its FASTQ reader; this restatement follows the parser statement by
// statement and is itself the checker for the device converter (isaac_gpu_fastq_to_bcl).
#include <stdint.h>
#include <string.h>
#include <string>

namespace
{
std::string g_fastqError;

inline bool isNewLine(char c) { return '\n' == c || '\r' == c; }
// findNotNewLine / findNewLine (FastqReader.cpp:103-118)
inline const char *findNotNewLine(const char *it, const char *end) { while (it != end && isNewLine(*it)) ++it; return it; }
inline const char *findNewLine(const char *it, const char *end) { while (it != end && !isNewLine(*it)) ++it; return it; }

// oligo::getTranslator(true, INCORRECT_FASTQ_BASE) (include/oligo/Nucleotides.hh:41-59, FastqReader.hh:86)
enum { INVALID_OLIGO = 4, INCORRECT_FASTQ_BASE = 5 };
inline unsigned translate(unsigned char c)
{
    switch (c)
    {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    case 'n': case 'N': return INVALID_OLIGO;
    default: return INCORRECT_FASTQ_BASE;
    }
}
} // namespace

extern "C" {

const char *oracle_fastq_last_error() { return g_fastqError.c_str(); }

// Error codes (what the reference throws): 1 FastqFormatException, 2 common::IoException (read length), 0 ok.
// `final`: the buffer ends at the end of the file; otherwise a record that is not followed by a newline inside the buffer
// is left for the next call (consumed_out = its first byte), as FastqReader::fetchMore would extend the buffer.
// The BCL bytes of cluster k go to bcl_out + k * cluster_stride (the caller adds the read's offset inside the cluster).
int oracle_fastq_to_bcl(const char *text, uint64_t n_bytes, uint32_t read_length, int allow_variable_length, int final,
                        uint8_t *bcl_out, uint64_t cluster_stride, uint32_t max_clusters,
                        uint32_t *n_clusters_out, uint64_t *consumed_out, uint64_t *error_offset_out)
{
    const char *const begin = text, *const end = text + n_bytes;
    const char *endIt = begin;          // FastqReader::endIt_: end of the previous record
    uint32_t clusters = 0;
    *n_clusters_out = 0; *consumed_out = 0; if (error_offset_out) *error_offset_out = 0;
#define FASTQ_FAIL(code, where, what) do { g_fastqError = what; if (error_offset_out) *error_offset_out = uint64_t((where) - begin); *n_clusters_out = clusters; return code; } while (0)
    while (clusters < max_clusters)
    {
        // findHeader (:120-163)
        const char *headerBegin = findNotNewLine(endIt, end);
        if (end == headerBegin) { endIt = end; break; }                          // no more records
        const char *headerEnd = findNewLine(headerBegin, end);
        if (end == headerEnd) { if (!final) break; FASTQ_FAIL(1, headerEnd, "Fastq file end while reading the header line"); }
        // findSequence (:165-213)
        const char *baseCallsBegin = findNotNewLine(headerEnd, end);
        if (end == baseCallsBegin) { if (!final) break; FASTQ_FAIL(1, baseCallsBegin, "Fastq file end while looking for sequence start"); }
        bool zeroLengthRead = false;
        const char *baseCallsEnd;
        if ('+' == *baseCallsBegin) { zeroLengthRead = true; baseCallsEnd = baseCallsBegin; }
        else baseCallsEnd = findNewLine(baseCallsBegin, end);
        if (end == baseCallsEnd) { if (!final) break; FASTQ_FAIL(1, baseCallsEnd, "Fastq file end while reading the sequence line"); }
        // findQScores (:215-263)
        const char *qScoresBegin = findNotNewLine(baseCallsEnd, end);
        if (end == qScoresBegin) { if (!final) break; FASTQ_FAIL(1, qScoresBegin, "Fastq file end while looking for + sign"); }
        if ('+' != *qScoresBegin) FASTQ_FAIL(1, qScoresBegin, "+ sign not found where expected");
        qScoresBegin = findNewLine(qScoresBegin, end);
        qScoresBegin = findNotNewLine(qScoresBegin, end);
        if (end == qScoresBegin)
        {
            if (!final) break;
            // the reference throws here even for a zero-length read at the very end of the file
            FASTQ_FAIL(1, qScoresBegin, "Fastq file end while looking for qscores");
        }
        // findQScoresEnd (:265-283)
        const char *recordEnd;
        if (zeroLengthRead) recordEnd = qScoresBegin;
        else
        {
            recordEnd = findNewLine(qScoresBegin, end);
            if (end == recordEnd && !final) break;                                   // fetchMore would continue the line
        }
        // extractBcl (FastqReader.hh:144-210), all cycles of the read used
        uint8_t *it = bcl_out + uint64_t(clusters) * cluster_stride;
        const char *baseCallsIt = baseCallsBegin, *qScoresIt = qScoresBegin;
        uint32_t extracted = 0;
        for (; recordEnd != qScoresIt && extracted < read_length; ++baseCallsIt, ++qScoresIt)
        {
            const unsigned baseValue = translate((unsigned char)*baseCallsIt);
            if (INVALID_OLIGO == baseValue) *it = 0;
            else if (INCORRECT_FASTQ_BASE == baseValue) FASTQ_FAIL(1, baseCallsIt, "Invalid oligo found");
            else
            {
                const unsigned char baseQuality = (unsigned char)(*qScoresIt - 33);
                if ((1 << 6) <= baseQuality) FASTQ_FAIL(1, baseCallsIt, "Invalid quality found. Base quality scores [0-63] supported only.");
                *it = uint8_t(baseValue | (baseQuality << 2));
            }
            ++it; ++extracted;
        }
        if (!allow_variable_length) { if (read_length != extracted) FASTQ_FAIL(2, headerBegin, "Read length is different from expected"); }
        else for (; extracted < read_length; ++extracted) *it++ = 0;
        ++clusters;
        endIt = recordEnd;
    }
#undef FASTQ_FAIL
    *n_clusters_out = clusters;
    // what the next call has to see again: everything from the end of the last complete record
    *consumed_out = uint64_t(endIt - begin);
    return 0;
}

} // extern "C"

// FastqSeedSource: tileClustersMax_ (FastqDataSource.cpp:82-84) and the tile breakdown of discoverTiles (:153-173): full tiles, then one
// partial tile for what is left; a load that is a whole number of tiles ends without a partial one
#include <algorithm>
#include <utility>
#include <vector>
namespace oracle
{
unsigned fastqTileClustersMax(unsigned clustersAtATimeMax, unsigned seedCount)
{
    const unsigned seedBound = 40000000 / seedCount;
    return (clustersAtATimeMax && clustersAtATimeMax < seedBound) ? clustersAtATimeMax : seedBound;
}
void fastqDiscoverTiles(unsigned clustersLoaded, unsigned tileClustersMax, unsigned &currentTile, std::vector<std::pair<unsigned, unsigned> > &loadedTiles)
{
    const unsigned fullTiles = clustersLoaded / tileClustersMax, rest = clustersLoaded % tileClustersMax;
    for (unsigned t = 0; t < fullTiles; ++t) loadedTiles.push_back(std::make_pair(currentTile++, tileClustersMax));
    if (rest) loadedTiles.push_back(std::make_pair(currentTile++, rest));
}
} // namespace oracle
