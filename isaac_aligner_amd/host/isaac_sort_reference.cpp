// isaac-sort-reference on one MI355X: what the reference's bash/bin/isaac-sort-reference drives through make/reference/SortReference.mk --
// printContigs (lib/reference/ContigsPrinter.cpp:46-141: the contig table of the FASTA file), sortReference once per mask
// (lib/reference/ReferenceSorter.cpp), mergeReferences, findNeighbors (lib/reference/NeighborsFinder.cpp) -- as one program on
// include/isaac_gpu.h: the contigs are read as reference::loadContig reads them, the 32-mer table with its neighbour flags is built on the
// device (isaac_gpu_build_index) and written as <genome>-32mer-6bit-ABCD-NN.dat + sorted-reference.xml (isaac_gpu_save_sorted_reference).
// Not written: genome-neighbors.1bpb and repeats-<threshold>.1bpb (extractNeighbors; inputs of the reference's reports, not of isaac-align).
#include "isaac_gpu.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace
{

const char *VERSION = "isaac_aligner_amd-0.3";

// RFC 1321, for the M5 attribute of a contig (common::MD5Sum in the reference)
class Md5
{
public:
    Md5() { clear(); }
    void clear() { a_ = 0x67452301u; b_ = 0xefcdab89u; c_ = 0x98badcfeu; d_ = 0x10325476u; length_ = 0; fill_ = 0; }
    void update(const char *data, size_t n)
    {
        length_ += n;
        while (n)
        {
            const size_t take = std::min<size_t>(n, 64 - fill_);
            std::memcpy(buffer_ + fill_, data, take);
            fill_ += take; data += take; n -= take;
            if (64 == fill_) { block(buffer_); fill_ = 0; }
        }
    }
    std::string hex()
    {
        const uint64_t bits = length_ * 8;
        const unsigned char one = 0x80, zero = 0;
        update(reinterpret_cast<const char *>(&one), 1);
        while (56 != fill_) update(reinterpret_cast<const char *>(&zero), 1);
        unsigned char tail[8];
        for (int i = 0; i < 8; ++i) tail[i] = static_cast<unsigned char>(bits >> (8 * i));
        update(reinterpret_cast<const char *>(tail), 8);
        char text[33];
        const uint32_t words[4] = { a_, b_, c_, d_ };
        for (int w = 0; w < 4; ++w) for (int i = 0; i < 4; ++i) std::snprintf(text + 8 * w + 2 * i, 3, "%02x", (words[w] >> (8 * i)) & 0xffu);
        return std::string(text, 32);
    }
private:
    static uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
    void block(const unsigned char *p)
    {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
        static const int S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                   4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = uint32_t(p[4 * i]) | uint32_t(p[4 * i + 1]) << 8 | uint32_t(p[4 * i + 2]) << 16 | uint32_t(p[4 * i + 3]) << 24;
        uint32_t a = a_, b = b_, c = c_, d = d_;
        for (int i = 0; i < 64; ++i)
        {
            uint32_t f; int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            const uint32_t next = b + rotl(a + f + K[i] + m[g], S[i]);
            a = d; d = c; c = b; b = next;
        }
        a_ += a; b_ += b; c_ += c; d_ += d;
    }
    uint32_t a_, b_, c_, d_; uint64_t length_; unsigned char buffer_[64]; size_t fill_;
};

void usage(const std::string &defaultOutput)
{
    std::cout <<
        "Usage: isaac-sort-reference [options]\n"
        "Options:\n"
        "  -g [ --genome-file ] arg                 Path to fasta file containing the reference contigs\n"
        "  -h [ --help ]                            Print this message\n"
        "  -j [ --jobs ] arg (=1)                   Accepted; the work is done on the GPU\n"
        "  -o [ --output-directory ] arg (" << defaultOutput << ") Location where the results are stored\n"
        "  -q [ --quiet ]                           Avoid excessive logging\n"
        "  -p [ --no-parallel-sort ]                Accepted; no effect\n"
        "  -s [ --seed-length ] arg (=32)           Length of the k-mer. Only 32-mer sorted references are built by this program\n"
        "  -t [ --repeat-threshold ] arg (=1000)    Repeat cutoff after which individual kmer positions are not stored\n"
        "  -v [ --version ]                         Only print version information\n"
        "  -w [ --mask-width ] arg (=6)             Number of high order bits that split the table into files. Only 6 (64 files)\n"
        "  --dont-annotate                          Don't search for neighbors\n"
        "  --annotate                               Force neighbor search (the default)\n"
        "  --device arg (=0)                        HIP device\n";
}

std::string absolutePath(const std::string &path)
{
    if (!path.empty() && '/' == path[0]) return path;
    char cwd[4096];
    if (!::getcwd(cwd, sizeof(cwd))) throw std::runtime_error("getcwd failed");
    return std::string(cwd) + "/" + path;
}

void makeDirectories(const std::string &path)
{
    for (size_t at = 1; at <= path.size(); ++at)
        if (at == path.size() || '/' == path[at])
        {
            const std::string prefix = path.substr(0, at);
            if (::mkdir(prefix.c_str(), 0777) && EEXIST != errno) throw std::runtime_error("Failed to create directory " + prefix + ": " + std::strerror(errno));
        }
}

void setText(char *to, size_t capacity, const std::string &text, const char *what)
{
    if (text.size() >= capacity) throw std::runtime_error(std::string(what) + " is too long: " + text);
    std::memcpy(to, text.c_str(), text.size() + 1);
}

} // namespace

int main(int argc, char **argv)
{
    std::string genomeFile, outputDirectory;
    {
        char date[32]; const std::time_t now = std::time(0); std::strftime(date, sizeof(date), "%Y%m%d", std::localtime(&now));
        outputDirectory = std::string("./iSAACIndex.") + date;
    }
    const std::string defaultOutput = outputDirectory;
    unsigned repeatThreshold = 1000, seedLength = 32, maskWidth = 6;
    int device = 0;
    bool annotate = true, quiet = false;
    try
    {
        for (int i = 1; i < argc; ++i)
        {
            const std::string param = argv[i];
            const auto value = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("ERROR: " + param + " needs an argument"); return argv[++i]; };
            if (param == "--mask-width" || param == "-w") maskWidth = unsigned(std::atoi(value().c_str()));
            else if (param == "--genome-file" || param == "-g") genomeFile = absolutePath(value());
            else if (param == "--dont-annotate") annotate = false;
            else if (param == "--annotate") annotate = true;
            else if (param == "--dry-run" || param == "-n") { std::cerr << "ERROR: --dry-run: there are no commands to print, the table is built in this process" << std::endl; return 2; }
            else if (param == "--output-directory" || param == "-o") outputDirectory = value();
            else if (param == "--repeat-threshold" || param == "-t") repeatThreshold = unsigned(std::atoi(value().c_str()));
            else if (param == "--jobs" || param == "-j") value();
            else if (param == "--no-paralle-sort" || param == "--no-parallel-sort" || param == "-p") {}
            else if (param == "--seed-length" || param == "-s") seedLength = unsigned(std::atoi(value().c_str()));
            else if (param == "--help" || param == "-h") { usage(defaultOutput); return 1; }
            else if (param == "--version" || param == "-v") { std::cout << VERSION << std::endl; return 1; }
            else if (param == "--quiet" || param == "-q") quiet = true;
            else if (param == "--device") device = std::atoi(value().c_str());
            else { std::cerr << "ERROR: unrecognized argument: " << param << std::endl; return 2; }
        }
        if (outputDirectory.empty() || genomeFile.empty()) { usage(defaultOutput); std::cerr << "ERROR: --output-directory and --genome-file arguments are mandatory" << std::endl; return 2; }
        struct stat st;
        if (::stat(genomeFile.c_str(), &st)) { std::cout << "ERROR: File not found: '" << genomeFile << "'" << std::endl; return 2; }
        if (16 != seedLength && 32 != seedLength && 64 != seedLength) { usage(defaultOutput); std::cerr << "ERROR: --seed-length must be 16, 32 or 64" << std::endl; return 2; }
        if (32 != seedLength) { std::cerr << "ERROR: --seed-length " << seedLength << ": this program builds 32-mer references only" << std::endl; return 2; }
        if (6 != maskWidth) { std::cerr << "ERROR: --mask-width " << maskWidth << ": this program writes 64 mask files (--mask-width 6) only" << std::endl; return 2; }
        if (!repeatThreshold) { std::cerr << "ERROR: --repeat-threshold must be positive" << std::endl; return 2; }
        outputDirectory = absolutePath(outputDirectory);
        makeDirectories(outputDirectory);

        // ---- printContigs (ContigsPrinter::run) and reference::loadContig in one pass over the file
        std::ifstream is(genomeFile.c_str(), std::ios::binary);
        if (!is) throw std::runtime_error("Failed to open reference file " + genomeFile);
        std::vector<isaac_reference_contig> contigs;
        std::string bases, line, upper;
        std::vector<uint64_t> offsets(1, 0);
        Md5 md5;
        uint64_t streamPos = 0, genomicStart = 0;
        bool haveContig = false;
        const auto finish = [&](uint64_t byteEnd)
        {
            isaac_reference_contig &c = contigs.back();
            c.size = byteEnd - c.offset;
            setText(c.bam_m5, sizeof(c.bam_m5), md5.hex(), "M5");
            genomicStart += c.total_bases;
            offsets.push_back(bases.size());
            if (bases.size() - offsets[offsets.size() - 2] != c.total_bases)
                throw std::runtime_error("Contig " + std::string(c.name) + " has characters that are not letters: " + std::to_string(c.total_bases) + " characters, " +
                                         std::to_string(bases.size() - offsets[offsets.size() - 2]) + " bases (the aligner's contig loader would refuse it)");
        };
        while (std::getline(is, line))
        {
            const uint64_t lineStart = streamPos;
            streamPos += line.size() + (is.eof() ? 0 : 1);
            if (!line.empty() && '>' == line[0])
            {
                if (haveContig) finish(lineStart);
                isaac_reference_contig c; std::memset(&c, 0, sizeof(c));
                const std::string name = line.substr(1, line.find_first_of(" \t\r") == std::string::npos ? std::string::npos : line.find_first_of(" \t\r") - 1);
                setText(c.name, sizeof(c.name), name, "contig name");
                setText(c.file, sizeof(c.file), genomeFile, "genome file path");
                c.genomic_position = genomicStart; c.offset = streamPos; c.index = c.karyotype_index = uint32_t(contigs.size());
                contigs.push_back(c);
                haveContig = true;
                md5.clear();
            }
            else if (haveContig)
            {
                isaac_reference_contig &c = contigs.back();
                upper.clear();
                for (const char ch : line)
                {
                    if ('\r' != ch) ++c.total_bases;                                        // "ignore untranslated '\r'"
                    const unsigned char u = static_cast<unsigned char>(ch);
                    const char up = char(std::toupper(u));
                    if ('A' == up || 'C' == up || 'G' == up || 'T' == up) ++c.acgt_bases;
                    if (!std::isspace(u)) upper.push_back(up);                              // MD5 with format characters stripped out
                    if (std::isalpha(u)) bases.push_back(('A' == up || 'C' == up || 'G' == up || 'T' == up) ? up : 'N');     // reference::loadContig
                }
                md5.update(upper.data(), upper.size());
            }
        }
        if (haveContig) finish(streamPos);
        if (contigs.empty()) throw std::runtime_error("No contigs in " + genomeFile);
        if (!quiet) for (const isaac_reference_contig &c : contigs)
            std::cerr << "isaac-sort-reference: contig " << c.name << ": " << c.total_bases << " bases (" << c.acgt_bases << " ACGT) at byte " << c.offset << ", M5 " << c.bam_m5 << std::endl;

        // ---- the table
        isaac_params params;
        if (isaac_gpu_default_params(100, 100, &params)) throw std::runtime_error(isaac_gpu_params_last_error());
        isaac_gpu_ctx *ctx = 0;
        const auto check = [](int rc, const char *what) { if (rc) throw std::runtime_error(std::string(what) + ": error " + std::to_string(rc) + ": " + isaac_gpu_last_error()); };
        check(isaac_gpu_create(device, &params, 0, &ctx), "isaac_gpu_create");
        check(isaac_gpu_load_contigs(ctx, bases.data(), offsets.data(), uint32_t(contigs.size())), "isaac_gpu_load_contigs");
        uint64_t nEntries = 0;
        check(isaac_gpu_build_index(ctx, repeatThreshold, annotate ? 1 : 0, &nEntries), "isaac_gpu_build_index");
        const std::string genomeName = genomeFile.substr(genomeFile.rfind('/') + 1);
        check(isaac_gpu_save_sorted_reference(ctx, outputDirectory.c_str(), genomeName.c_str(), contigs.data(), uint32_t(contigs.size())), "isaac_gpu_save_sorted_reference");
        isaac_gpu_destroy(ctx);
        if (!quiet) std::cerr << "isaac-sort-reference: " << nEntries << " entries in " << outputDirectory << "/" << genomeName << "-32mer-6bit-ABCD-*.dat, " << outputDirectory << "/sorted-reference.xml" << std::endl;
        return 0;
    }
    catch (const std::exception &e)
    {
        std::cerr << "isaac-sort-reference: " << e.what() << std::endl;
        return 2;
    }
}
