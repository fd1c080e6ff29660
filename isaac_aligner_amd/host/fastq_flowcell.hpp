// A FASTQ flowcell as options::alignOptions::FastqFlowcell sees it (lib/options/alignOptions/FastqFlowcell.cpp:36-330): the lanes that have
// lane<N>_read<R>.fastq[.gz] files (lib/flowcell/FastqLayout.cpp:31-53), the read lengths and the flowcell id of the first record, the
// --use-bases-mask expansion (include/options/UseBasesMaskGrammar.hh:49-125, lib/options/alignOptions/UseBasesMaskOption.cpp:63-171), and
// a reader that hands the text of a lane file over in pieces (gzip members inflated on the way).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <zlib.h>

namespace isaac_host
{

struct FastqLane { unsigned lane = 0; std::string readPath[2]; };          // an empty path: the lane has no such read

struct FastqFlowcell
{
    std::string baseCallsDirectory, flowcellId;          // the id is never empty ("unknown-flowcell")
    bool compressed = false;
    std::vector<FastqLane> lanes;
    unsigned fileReadLength[2] = { 0, 0 };               // bases per record in the files
    unsigned readLength[2] = { 0, 0 };                   // cycles used: the y-prefix of the read's mask
    unsigned nReads = 0;

    // throws InvalidOption / std::runtime_error as the reference does for missing lanes, mismatching ids and lengths
    static FastqFlowcell discover(const std::string &baseCallsDirectory, bool compressed, unsigned laneNumberMax, const std::string &useBasesMask, bool allowVariableLength);
};

// expands one use-bases-mask for reads of the given lengths: one string of y / n / i per read
std::vector<std::string> expandUseBasesMask(const std::vector<unsigned> &readLengths, const std::string &useBasesMask, const std::string &baseCallsDirectory);

// sequential reader of a lane file: plain or gzip (concatenated members included)
class FastqFileReader
{
public:
    FastqFileReader(const std::string &path, bool compressed);
    ~FastqFileReader();
    FastqFileReader(const FastqFileReader &) = delete;
    FastqFileReader &operator=(const FastqFileReader &) = delete;
    // appends up to `want` bytes of text to `to`; returns the number appended, 0 at the end of the file
    size_t read(std::vector<char> &to, size_t want);
    // the same into memory of the caller's: up to `want` bytes at `to`
    size_t readInto(char *to, size_t want);
    bool atEnd() const { return eof_; }
    const std::string &path() const { return path_; }
private:
    std::string path_;
    bool compressed_, eof_ = false, streamOpen_ = false, seekable_ = true;   // seekable_: a regular file (read at positions, several threads side by side)
    size_t readThreads_ = 8;
    uint64_t position_ = 0;                   // of a plain file: where the next piece begins
    std::FILE *file_ = 0;
    z_stream z_;
    std::vector<unsigned char> in_;
};

} // namespace isaac_host
