#include "fastq_flowcell.hpp"
#include "align_options.hpp"

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <iostream>
#include <stdexcept>
#include <sys/stat.h>

namespace isaac_host
{
namespace
{

bool exists(const std::string &path) { struct stat st; return 0 == ::stat(path.c_str(), &st); }

// the first record of a lane file: its header line and the length of its sequence line; false for a file without data
bool firstRecord(const std::string &path, bool compressed, std::string &header, unsigned &sequenceLength)
{
    // The flowcell is looked into before it is read (as the reference does: FastqFlowcell.cpp:145-182 opens every lane for its first header), so a lane has to be
    // a file that can be opened twice: what is peeked from a pipe or /dev/stdin is lost to the reader proper, and records would go missing without an error.
    struct stat st;
    if (0 == ::stat(path.c_str(), &st) && !S_ISREG(st.st_mode)) throw std::runtime_error("Fastq lane is not a regular file (pipes and devices are not supported): " + path);
    FastqFileReader reader(path, compressed);
    std::vector<char> text;
    size_t lineEnds = 0;
    // two line ends are enough; records are tiny, but a header may be long
    while (lineEnds < 2 && reader.read(text, 1 << 16)) lineEnds = size_t(std::count(text.begin(), text.end(), '\n'));
    if (text.empty()) return false;
    const auto isEol = [](char c) { return c == '\n' || c == '\r'; };
    std::vector<char>::const_iterator at = text.begin();
    while (at != text.end() && isEol(*at)) ++at;
    const std::vector<char>::const_iterator headerEnd = std::find_if(at, text.cend(), isEol);
    header.assign(at, headerEnd);
    at = headerEnd;
    while (at != text.end() && isEol(*at)) ++at;
    sequenceLength = unsigned(std::find_if(at, text.cend(), isEol) - at);
    return !header.empty();
}

// CasavaFastqParser::parseFlowcellId (FastqFlowcell.cpp:45-81): the third field of "@instrument:run:flowcell:lane..." (':' or ' ' separate)
std::string parseFlowcellId(const std::string &header)
{
    if ('@' != header[0]) throw std::runtime_error("Fastq header must begin with @: " + header);
    const char *delimiters = ": ";
    size_t at = header.find_first_of(delimiters, 1);
    if (std::string::npos == at) return "";
    at = header.find_first_of(delimiters, at + 1);
    if (std::string::npos == at) return "";
    const size_t end = header.find_first_of(delimiters, at + 1);
    return header.substr(at + 1, std::string::npos == end ? std::string::npos : end - at - 1);
}

} // namespace

std::vector<std::string> expandUseBasesMask(const std::vector<unsigned> &readLengths, const std::string &useBasesMask, const std::string &baseCallsDirectory)
{
    const auto fail = [&](size_t at) { return InvalidOption("\n   *** Could not parse the use-bases-mask '" + useBasesMask + "' for '" + baseCallsDirectory + "' at: " + useBasesMask.substr(at) + " ***\n"); };
    std::vector<std::string> result;
    size_t at = 0;
    for (;;)
    {
        const unsigned length = result.size() < readLengths.size() ? readLengths[result.size()] : 0;
        std::string mask, afterStar;          // one '*' per read: what follows it is kept apart until the read's length is known
        bool star = false; char starChar = 0;
        while (at < useBasesMask.size() && ',' != useBasesMask[at])
        {
            const char c = char(std::tolower(static_cast<unsigned char>(useBasesMask[at])));
            if ('y' != c && 'n' != c && 'i' != c) throw fail(at);
            ++at;
            std::string &to = star ? afterStar : mask;
            to.push_back(c);
            if (at < useBasesMask.size() && std::isdigit(static_cast<unsigned char>(useBasesMask[at])))
            {
                size_t digits = at; unsigned long count = 0;
                while (digits < useBasesMask.size() && std::isdigit(static_cast<unsigned char>(useBasesMask[digits]))) count = count * 10 + (useBasesMask[digits++] - '0');
                if (!count) throw fail(at);
                to.append(count - 1, c);
                at = digits;
            }
            else if (at < useBasesMask.size() && '*' == useBasesMask[at])
            {
                if (star) throw fail(at);
                star = true; starChar = c; ++at;
            }
        }
        if (star && length > mask.size() + afterStar.size()) mask.append(length - mask.size() - afterStar.size(), starChar);
        result.push_back(mask + afterStar);
        if (at >= useBasesMask.size()) break;
        ++at;                                   // ','
    }
    if (result.size() != readLengths.size())
        throw InvalidOption("\n   *** use-bases-mask '" + useBasesMask + "' is incompatible with number of reads (" + std::to_string(readLengths.size()) + ") in " + baseCallsDirectory + " ***\n");
    return result;
}

FastqFlowcell FastqFlowcell::discover(const std::string &baseCallsDirectory, bool compressed, unsigned laneNumberMax, const std::string &useBasesMaskOption, bool allowVariableLength)
{
    FastqFlowcell fc;
    fc.baseCallsDirectory = baseCallsDirectory; fc.compressed = compressed;
    // FastqFlowcell::findFastqPathPairs
    std::vector<FastqLane> found;
    for (unsigned lane = 1; lane <= laneNumberMax; ++lane)
    {
        FastqLane l; l.lane = lane;
        for (unsigned read = 0; read < 2; ++read)
        {
            const std::string path = baseCallsDirectory + "/lane" + std::to_string(lane) + "_read" + std::to_string(read + 1) + (compressed ? ".fastq.gz" : ".fastq");
            if (exists(path)) l.readPath[read] = path;
        }
        if (!l.readPath[0].empty() || !l.readPath[1].empty()) found.push_back(l);
    }
    if (found.empty()) throw InvalidOption("\n   *** Could not find any fastq lanes in: " + baseCallsDirectory + " ***\n");
    // FastqFlowcell::parseFastqFlowcellInfo: the first lane with data decides, the others must agree
    bool ready = false;
    for (const FastqLane &l : found)
    {
        unsigned lengths[2] = { 0, 0 };
        std::string id;
        for (unsigned read = 0; read < 2; ++read)
        {
            if (l.readPath[read].empty()) continue;
            std::string header;
            if (!firstRecord(l.readPath[read], compressed, header, lengths[read])) { lengths[read] = 0; continue; }
            const std::string readId = parseFlowcellId(header);
            if (0 == read || id.empty()) id = readId;
            else if (id != readId) throw std::runtime_error("Flowcell ID mismatch between fastq reads " + id + " vs " + readId + ", " + l.readPath[0] + ", " + l.readPath[1]);
        }
        if (!lengths[0] && !lengths[1]) { std::cerr << "WARNING: Skipping lane " << l.lane << " due to read length 0" << std::endl; continue; }
        if (!ready) { fc.fileReadLength[0] = lengths[0]; fc.fileReadLength[1] = lengths[1]; fc.flowcellId = id; ready = true; }
        else
        {
            if (!allowVariableLength && (lengths[0] != fc.fileReadLength[0] || lengths[1] != fc.fileReadLength[1]))
                throw std::runtime_error("Read lengths mismatch between lanes of the same flowcell " + baseCallsDirectory + ": lane " + std::to_string(l.lane));
            if (id != fc.flowcellId) std::cerr << "WARNING: Flowcell id mismatch across the lanes of the same flowcell " << id << " vs " << fc.flowcellId << std::endl;
        }
        fc.lanes.push_back(l);
    }
    if (fc.lanes.empty()) throw InvalidOption("\n   *** " + baseCallsDirectory + " has no data. Use --allow-empty-flowcell to avoid the failure. ***\n");
    if (fc.flowcellId.empty()) fc.flowcellId = "unknown-flowcell";                          // AlignOptions.cpp:1226
    // createFilteredFlowcell: the reads that exist, the default mask, the cycles in use
    std::vector<unsigned> readLengths;
    for (unsigned read = 0; read < 2; ++read) if (fc.fileReadLength[read]) readLengths.push_back(fc.fileReadLength[read]);
    if (readLengths.size() == 1 && !fc.fileReadLength[0])
        throw InvalidOption("\n   *** " + baseCallsDirectory + ": lanes with a second read only are not supported by this host ***\n");
    const std::string useBasesMask = "default" != useBasesMaskOption ? useBasesMaskOption : 1 == readLengths.size() ? "y*n" : "y*n,y*n";
    const std::vector<std::string> masks = expandUseBasesMask(readLengths, useBasesMask, baseCallsDirectory);
    fc.nReads = unsigned(readLengths.size());
    for (unsigned read = 0; read < fc.nReads; ++read)
    {
        const std::string &mask = masks[read];
        const size_t used = mask.find_first_not_of('y');
        const size_t cycles = std::string::npos == used ? mask.size() : used;
        if (std::string::npos != mask.find('i')) throw InvalidOption("\n   *** use-bases-mask '" + useBasesMask + "': index cycles are not supported by this host ***\n");
        if (!cycles || std::string::npos != mask.find('y', cycles))
            throw InvalidOption("\n   *** use-bases-mask '" + useBasesMask + "': this host uses the cycles of a read from its first one up to the first masked one; '" + mask + "' is not of that form ***\n");
        if (cycles > readLengths[read]) throw InvalidOption("\n   *** use-bases-mask '" + useBasesMask + "' asks for more cycles than read " + std::to_string(read + 1) + " has ***\n");
        if (cycles < 32) throw InvalidOption("\n   *** read " + std::to_string(read + 1) + " is too short: " + std::to_string(cycles) + " cycle < 32 in " + baseCallsDirectory + " ***\n");
        fc.readLength[read] = unsigned(cycles);
    }
    return fc;
}

FastqFileReader::FastqFileReader(const std::string &path, bool compressed) : path_(path), compressed_(compressed)
{
    file_ = std::fopen(path.c_str(), "rb");
    if (!file_) throw std::runtime_error("Failed to open file " + path + ": " + std::strerror(errno));
    std::memset(&z_, 0, sizeof(z_));
    if (compressed_) in_.resize(1 << 20);
    struct stat st;
    seekable_ = 0 == ::fstat(fileno(file_), &st) && S_ISREG(st.st_mode);
    if (const char *e = std::getenv("ISAAC_ALIGN_READ_THREADS")) readThreads_ = size_t(std::max(1, std::atoi(e)));
}

FastqFileReader::~FastqFileReader()
{
    if (streamOpen_) inflateEnd(&z_);
    if (file_) std::fclose(file_);
}

size_t FastqFileReader::readInto(char *out, size_t want)
{
    if (eof_ || !want) return 0;
    size_t got = 0;
    if (!compressed_)
    {
        // plain text: a large piece is fetched by a few threads side by side (one thread copies 8 GB/s out of the page cache; the two files of a
        // lane are 660 bytes per pair)
        const int fd = fileno(file_);
        if (!seekable_)
        {   // a pipe, a process substitution, /dev/stdin: no positions to read from side by side
            while (got < want)
            {
                const size_t r = std::fread(out + got, 1, want - got, file_);
                if (!r) { if (std::ferror(file_)) throw std::runtime_error("Failed to read " + path_ + ": " + std::strerror(errno)); break; }
                got += r;
            }
            if (got < want) eof_ = true;
            return got;
        }
        const size_t threads = want >= (size_t(16) << 20) ? readThreads_ : 1, share = (want + threads - 1) / threads;
        std::vector<size_t> gotPart(threads, 0);
        std::vector<int> failed(threads, 0);
        const auto part = [&](size_t t)
        {
            const size_t first = t * share, n = first < want ? std::min(share, want - first) : 0;
            size_t done = 0;
            while (done < n)
            {
                const ssize_t r = ::pread(fd, out + first + done, n - done, off_t(position_ + first + done));
                if (r < 0) { if (EINTR == errno) continue; failed[t] = errno ? errno : EIO; break; }
                if (!r) break;
                done += size_t(r);
            }
            gotPart[t] = done;
        };
        std::vector<std::thread> workers;
        for (size_t t = 1; t < threads; ++t) workers.emplace_back(part, t);
        part(0);
        for (std::thread &w : workers) w.join();
        for (size_t t = 0; t < threads; ++t)
        {
            if (failed[t]) throw std::runtime_error("Failed to read " + path_ + ": " + std::strerror(failed[t]));
            got += gotPart[t];
            if (gotPart[t] < (t * share < want ? std::min(share, want - t * share) : 0)) break;       // the file ends inside this part
        }
        position_ += got;
        if (got < want) eof_ = true;
    }
    else
    {
        while (got < want)
        {
            if (!z_.avail_in)
            {
                z_.next_in = in_.data();
                z_.avail_in = uInt(std::fread(in_.data(), 1, in_.size(), file_));
                if (!z_.avail_in)
                {
                    if (std::ferror(file_)) throw std::runtime_error("Failed to read " + path_);
                    if (streamOpen_) throw std::runtime_error("Unexpected end of compressed data in " + path_);
                    eof_ = true; break;
                }
            }
            if (!streamOpen_)
            {   // a new gzip member
                if (Z_OK != inflateInit2(&z_, 15 + 16)) throw std::runtime_error("inflateInit2 failed for " + path_);
                streamOpen_ = true;
            }
            z_.next_out = reinterpret_cast<Bytef *>(out + got);
            z_.avail_out = uInt(std::min<size_t>(want - got, 1u << 30));
            const uInt room = z_.avail_out;
            const int rc = inflate(&z_, Z_NO_FLUSH);
            got += room - z_.avail_out;
            if (Z_STREAM_END == rc)
            {   // the member is complete; another one may follow
                unsigned char *next = z_.next_in; const uInt left = z_.avail_in;
                inflateEnd(&z_); streamOpen_ = false;
                std::memset(&z_, 0, sizeof(z_));
                z_.next_in = next; z_.avail_in = left;
            }
            else if (Z_OK != rc && Z_BUF_ERROR != rc) throw std::runtime_error("Failed to decompress " + path_ + ": " + (z_.msg ? z_.msg : "zlib error"));
        }
    }
    return got;
}
size_t FastqFileReader::read(std::vector<char> &to, size_t want)
{
    if (eof_ || !want) return 0;
    const size_t before = to.size();
    to.resize(before + want);
    const size_t got = readInto(to.data() + before, want);
    to.resize(before + got);
    return got;
}

} // namespace isaac_host
