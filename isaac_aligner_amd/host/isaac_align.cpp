// isaac-align on one MI355X: the host the reference's bin/isaac-align.cpp + workflow::AlignWorkflow are for FASTQ data and one sample,
// written against include/isaac_gpu.h only.
//   options        options::AlignOptions                                        align_options.cpp
//   flowcells      options::alignOptions::FastqFlowcell, FastqSeedSource        fastq_flowcell.cpp; tiles by isaac_gpu_fastq_tiles
//   reference      reference::loadSortedReferenceXml, reference::loadContigs    (lib/reference/ContigLoader.cpp:29-66) + isaac_gpu_load_sorted_reference
//   find matches   workflow::alignWorkflow::FindMatchesTransition               isaac_gpu_fastq_to_bcl, isaac_gpu_find_matches (which contigs have matches)
//   select         workflow::alignWorkflow::SelectMatchesTransition             isaac_gpu_find_matches again (the matches are not kept: 1.3 ms per million
//                                                                               pairs against 400 bytes per pair), isaac_gpu_determine_tls per lane
//                                                                               (MatchSelector.cpp:395-412), isaac_gpu_select_n, isaac_gpu_compact_cigars,
//                  alignment::matchSelector::BinningFragmentStorage             isaac_gpu_bin_tile: every tile's clusters to the bins (one per contig) in host memory
//   build          build::Build                                                 per bin: isaac_gpu_bam_records (duplicates, realignment, order, records),
//                                                                               isaac_gpu_bgzf_deflate / _store, the blocks to sorted.bam, the records to the .bai
// --devices a,b,...: a worker (context + thread) per entry; loads, tiles and bins are dealt to them; one file comes out, in bin order.
#include "isaac_gpu.h"
#include "align_options.hpp"
#include "fastq_flowcell.hpp"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <sys/stat.h>
#include <sys/mman.h>
#include <sys/statvfs.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace isaac_host;

namespace
{

const char *VERSION = "isaac_aligner_amd-0.3";

void check(int rc, const char *what) { if (rc) throw std::runtime_error(std::string(what) + ": error " + std::to_string(rc) + ": " + isaac_gpu_last_error()); }
#define GPU(call) check((call), #call)

double seconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Stage
{
    const char *name; double start;
    explicit Stage(const char *n) : name(n), start(seconds()) { std::cerr << "isaac-align: " << name << std::endl; }
    ~Stage() { std::cerr << "isaac-align: " << name << " done in " << (seconds() - start) << " s" << std::endl; }
};

// device memory of the context
class DeviceMemory
{
public:
    DeviceMemory() {}
    DeviceMemory(isaac_gpu_ctx *ctx, uint64_t bytes) { reset(ctx, bytes); }
    ~DeviceMemory() { release(); }
    DeviceMemory(const DeviceMemory &) = delete;
    DeviceMemory &operator=(const DeviceMemory &) = delete;
    DeviceMemory(DeviceMemory &&other) noexcept : ctx_(other.ctx_), p_(other.p_), bytes_(other.bytes_) { other.p_ = 0; other.bytes_ = 0; }
    DeviceMemory &operator=(DeviceMemory &&other) noexcept { if (this != &other) { release(); ctx_ = other.ctx_; p_ = other.p_; bytes_ = other.bytes_; other.p_ = 0; other.bytes_ = 0; } return *this; }
    void reset(isaac_gpu_ctx *ctx, uint64_t bytes) { release(); ctx_ = ctx; bytes_ = bytes; GPU(isaac_gpu_malloc(ctx, std::max<uint64_t>(bytes, 64), &p_)); }
    void release() { if (p_) isaac_gpu_free(ctx_, p_); p_ = 0; bytes_ = 0; }
    template <typename T> T *as() const { return static_cast<T *>(p_); }
    uint64_t bytes() const { return bytes_; }
private:
    isaac_gpu_ctx *ctx_ = 0; void *p_ = 0; uint64_t bytes_ = 0;
};

void makeDirectories(const std::string &path)
{
    for (size_t at = 1; at <= path.size(); ++at)
        if (at == path.size() || '/' == path[at])
        {
            const std::string prefix = path.substr(0, at);
            if (::mkdir(prefix.c_str(), 0777) && EEXIST != errno) throw std::runtime_error("Failed to create directory " + prefix + ": " + std::strerror(errno));
        }
}
std::string directoryOf(const std::string &path) { const size_t slash = path.rfind('/'); return std::string::npos == slash ? "." : path.substr(0, slash); }
bool fileExists(const std::string &path) { struct stat st; return 0 == ::stat(path.c_str(), &st); }

// ---- the reference: sorted-reference.xml and the bases of its contigs ------------------------------------------------------------------
struct Reference
{
    std::vector<isaac_reference_contig> contigs;      // karyotype order: the order of the BAM header and of contig ids in the records
    std::unique_ptr<char[]> bases; uint64_t totalBases = 0; std::vector<uint64_t> offsets;
};

// the contigs of sorted-reference.xml in karyotype order, their places in the concatenated genome; no bases yet
Reference parseReference(const std::string &xmlPath)
{
    std::ifstream is(xmlPath.c_str(), std::ios::binary);
    if (!is) throw std::runtime_error("Failed to open sorted reference file " + xmlPath);
    std::stringstream text; text << is.rdbuf();
    const std::string xml = text.str();
    uint32_t nContigs = 0, nMasks = 0, version = 0;
    if (isaac_gpu_sorted_reference_parse(xml.data(), xml.size(), 0, 0, &nContigs, 0, 0, &nMasks, &version)) throw std::runtime_error(xmlPath + ": " + isaac_gpu_sorted_reference_last_error());
    Reference ref;
    ref.contigs.resize(nContigs);
    std::vector<isaac_reference_mask_file> masks(nMasks);
    if (isaac_gpu_sorted_reference_parse(xml.data(), xml.size(), ref.contigs.data(), nContigs, &nContigs, masks.data(), nMasks, &nMasks, &version))
        throw std::runtime_error(xmlPath + ": " + isaac_gpu_sorted_reference_last_error());
    std::sort(ref.contigs.begin(), ref.contigs.end(), [](const isaac_reference_contig &a, const isaac_reference_contig &b) { return a.karyotype_index < b.karyotype_index; });
    // reference::loadContig: the alphabetic characters from the contig's offset on, ACGT as they are (upper case), everything else N.
    // The contigs are read side by side, a thread each at a time (a human genome is three billion bytes to look at: a minute for one
    // thread pushing them one by one into a string, a second for the machine's cores translating them through a table into place).
    ref.offsets.push_back(0);
    for (const isaac_reference_contig &c : ref.contigs) ref.offsets.push_back(ref.offsets.back() + c.total_bases);
    ref.totalBases = ref.offsets.back();
    return ref;
}
Reference loadReference(const std::string &xmlPath)
{
    Reference ref = parseReference(xmlPath);
    ref.bases.reset(new char[ref.totalBases ? ref.totalBases : 1]);
    unsigned char translate[256];
    for (unsigned b = 0; b < 256; ++b)
    {
        const char u = char(std::toupper(int(b)));
        translate[b] = !std::isalpha(int(b)) ? 0 : ('A' == u || 'C' == u || 'G' == u || 'T' == u) ? static_cast<unsigned char>(u) : static_cast<unsigned char>('N');
    }
    std::atomic<size_t> next(0);
    std::mutex errorLock; std::string error;
    const auto work = [&]()
    {
        std::vector<char> buffer(size_t(4) << 20);
        for (size_t k = next++; k < ref.contigs.size(); k = next++)
            try
            {
                const isaac_reference_contig &c = ref.contigs[k];
                std::string path = c.file;
                if (!fileExists(path) && '/' != path[0] && fileExists(directoryOf(xmlPath) + "/" + path)) path = directoryOf(xmlPath) + "/" + path;
                std::ifstream fasta(path.c_str(), std::ios::binary);
                if (!fasta) throw std::runtime_error("Failed to open reference file " + path);
                if (!fasta.seekg(std::streamoff(c.offset))) throw std::runtime_error("Failed to reach offset " + std::to_string(c.offset) + " in reference file " + path);
                char *out = ref.bases.get() + ref.offsets[k];
                uint64_t have = 0;
                while (have < c.total_bases && fasta)
                {
                    fasta.read(buffer.data(), std::streamsize(buffer.size()));
                    const size_t got = size_t(fasta.gcount());
                    for (size_t i = 0; i < got && have < c.total_bases; ++i)
                    {
                        const unsigned char t = translate[static_cast<unsigned char>(buffer[i])];
                        out[have] = char(t); have += t ? 1 : 0;
                    }
                }
                if (have != c.total_bases) throw std::runtime_error("Failed to read " + std::to_string(c.total_bases) + " bases from reference file " + path + ": " + std::to_string(have));
            }
            catch (const std::exception &e) { std::lock_guard<std::mutex> hold(errorLock); if (error.empty()) error = e.what(); }
    };
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::max<size_t>(1, ref.contigs.size())); ++t) threads.emplace_back(work);
    for (std::thread &t : threads) t.join();
    if (!error.empty()) throw std::runtime_error(error);
    return ref;
}

// ---- the data: lanes, loads, tiles -------------------------------------------------------------------------------------------------------
struct Worker;
struct Load;
struct Tile
{
    unsigned lane = 0, number = 0, index = 0, clusters = 0;     // tile number within the lane (the read name), index over the run (the records)
    Load *load = 0; uint64_t firstCluster = 0;                    // where its BCL bytes are: cluster firstCluster of its load
    Worker *worker = 0;
    isaac_tls tls;
    std::string namePrefix, readGroup;
};

// --clusters-at-a-time clusters of one lane as FastqSeedSource loads them: the BCL bytes, on the device of the worker the load was dealt to while that
// has room, in host memory otherwise (back on the device for the time its tiles are selected); gone once the last of its tiles is binned
struct Load
{
    Worker *worker = 0;
    uint32_t clusters = 0; uint64_t bytes = 0;
    DeviceMemory dev; std::unique_ptr<uint8_t[]> host;
    std::vector<Tile *> tiles;
    unsigned tilesLeft = 0;
};

const size_t TEXT_CHUNK = size_t(64) << 20;          // text per conversion call
const size_t TEXT_SLACK = size_t(4) << 20;           // what may be left of a piece when the next one is taken: less than a record, unless records are longer than this
// where the load phase's time goes, summed over its threads: waiting for text, upload + conversion, first lookup
std::mutex g_timersLock;
double g_textWaitSeconds = 0, g_convertSeconds = 0, g_firstLookupSeconds = 0, g_textOpenSeconds = 0, g_loadMemorySeconds = 0, g_loadPlaceSeconds = 0, g_resolveSeconds = 0;
void addTime(double &timer, double seconds) { std::lock_guard<std::mutex> hold(g_timersLock); timer += seconds; }

// One read of one lane: the file's text in pieces of TEXT_CHUNK bytes, each in a page-locked buffer of its own, read ahead of the conversion by a
// thread of the stream's (the file is read while the device converts the piece before: round 4 read and converted in turn).  The incomplete record a
// conversion leaves at the end of a piece is moved in front of the next piece -- a few hundred bytes, the only copy the text sees on the host.
class TextStream
{
public:
    TextStream(const std::string &path, bool compressed) : reader_(path, compressed)
    {
        for (Slot &s : slots_) { void *p = 0; GPU(isaac_gpu_host_malloc(TEXT_SLACK + TEXT_CHUNK, &p)); s.base = static_cast<char *>(p); }
        thread_ = std::thread([this] { readAhead(); });
    }
    ~TextStream()
    {
        { std::lock_guard<std::mutex> hold(lock_); stop_ = true; }
        change_.notify_all();
        if (thread_.joinable()) thread_.join();
        for (Slot &s : slots_) if (s.base) isaac_gpu_host_free(s.base);
    }
    TextStream(const TextStream &) = delete;
    TextStream &operator=(const TextStream &) = delete;
    const std::string &path() const { return reader_.path(); }
    uint64_t consumedBytes = 0;                                   // of the uncompressed text, for error messages
    // the text at hand: [data(), data() + size()); final(): the file ends with it
    const char *data() { acquire(); return slots_[current_].base + slots_[current_].begin; }
    size_t size() { acquire(); return slots_[current_].end - slots_[current_].begin; }
    bool final() { acquire(); return slots_[current_].last; }
    void consume(size_t n) { slots_[current_].begin += n; consumedBytes += n; }
    // what is left of this piece goes in front of the next one, which becomes the text at hand
    void advance()
    {
        acquire();
        Slot &from = slots_[current_];
        if (from.last) return;
        const size_t tail = from.end - from.begin;
        if (tail > TEXT_SLACK) throw std::runtime_error(reader_.path() + ": a record of more than " + std::to_string(TEXT_SLACK) + " bytes");
        const size_t next = (current_ + 1) % SLOTS;
        wait(next);
        Slot &to = slots_[next];
        if (tail) std::memcpy(to.base + to.begin - tail, from.base + from.begin, tail);
        to.begin -= tail;
        { std::lock_guard<std::mutex> hold(lock_); from.ready = false; }
        change_.notify_all();
        current_ = next;
    }
private:
    static const size_t SLOTS = 3;
    struct Slot { char *base = 0; size_t begin = 0, end = 0; bool last = false, ready = false; };
    void wait(size_t k)
    {
        const double start = seconds();
        std::unique_lock<std::mutex> hold(lock_);
        change_.wait(hold, [&] { return slots_[k].ready || !error_.empty(); });
        if (!slots_[k].ready) throw std::runtime_error(error_);
        hold.unlock();
        addTime(g_textWaitSeconds, seconds() - start);
    }
    void acquire() { if (!acquired_) { wait(0); acquired_ = true; } }
    void readAhead()
    {
        try
        {
            for (size_t k = 0; ; k = (k + 1) % SLOTS)
            {
                {
                    std::unique_lock<std::mutex> hold(lock_);
                    change_.wait(hold, [&] { return !slots_[k].ready || stop_; });
                    if (stop_) return;
                }
                Slot &s = slots_[k];
                const size_t got = reader_.readInto(s.base + TEXT_SLACK, TEXT_CHUNK);
                const bool last = reader_.atEnd();
                { std::lock_guard<std::mutex> hold(lock_); s.begin = TEXT_SLACK; s.end = TEXT_SLACK + got; s.last = last; s.ready = true; }
                change_.notify_all();
                if (last) return;
            }
        }
        catch (const std::exception &e) { { std::lock_guard<std::mutex> hold(lock_); error_ = e.what(); } change_.notify_all(); }
    }
    FastqFileReader reader_;
    Slot slots_[SLOTS];
    size_t current_ = 0; bool acquired_ = false, stop_ = false;
    std::mutex lock_; std::condition_variable change_; std::thread thread_; std::string error_;
};

// ---- the bins: what BinningFragmentStorage keeps in files (lib/alignment/matchSelector/BinningFragmentStorage.cpp) kept in memory ----------
// The bins of the BAM stage, in file order: a run of small contigs, a contig, or a stretch of a contig too large for one bin (the reference cuts its
// contigs into bins by the match distribution, include/alignment/matchSelector/BinIndexMap.hh:44-104; here by the reads expected per base), and one
// for the templates without a position.  Every bin is sorted, filtered for duplicates and realigned by itself (lib/build/BinSorter.cpp).  A tile
// leaves one part in every bin it has records in (isaac_gpu_bin_tile_map): BCL bytes, records and CIGAR words of the clusters concerned, about
// 150 + 2 x read length bytes per read.  That is the run's memory model: HBM holds the table, the BCL bytes of the loads until their tiles are
// selected (host memory beyond what fits), one tile's scratch and -- in the build stage -- one bin at a time.  The bins' parts stay where
// isaac_gpu_bin_tile_map wrote them, a block of device memory per tile, for as long as the device has room beside what the build stage will want (a
// quarter of its memory is left alone); the tiles after that leave their parts in host memory.  A part on the device goes into its bin's BAM
// stage as it lies there, no copy in either direction.
struct BinPart
{
    const Tile *tile; uint64_t clusters, words, bytes;
    std::unique_ptr<uint8_t[]> data;                             // in host memory
    std::shared_ptr<DeviceMemory> block; uint64_t offset = 0;    // in a block of device memory (offset: inside the block) ...
    int place = -1;                                              // ... of this worker's device
    bool spilled = false; uint64_t fileOffset = 0;               // in the bin's file under --temp-directory
};
struct Bin
{
    std::mutex lock; std::vector<BinPart> parts; uint64_t bytes = 0, records = 0; uint64_t firstPosition = 0, endPosition = 0; /* ReferencePosition values */ bool unaligned = false;
    // what host memory has no room for (--memory-limit): a file of the bin's own, as BinningFragmentStorage keeps all of its bins; parts appended under the bin's lock
    int spillFd = -1; uint64_t spillBytes = 0; std::string spillPath;
    Bin() {}
    Bin(const Bin &) = delete;
    ~Bin() { closeSpill(); }
    void closeSpill() { if (spillFd >= 0) { ::close(spillFd); ::unlink(spillPath.c_str()); spillFd = -1; } }
};

uint64_t align64(uint64_t v) { return (v + 63) & ~uint64_t(63); }
uint64_t referencePosition(uint64_t contig, uint64_t position) { return (((contig + 1) << 40) | position) << 1; }       // reference::ReferencePosition::getValue()

// contigs of `lengths` (in the order of their ids) into bins of about binBases bases: binOfContig, the cuts inside contigs, and every bin's range
struct BinPlan { std::vector<uint32_t> binOfContig; std::vector<uint64_t> cuts; std::vector<std::pair<uint64_t, uint64_t> > ranges; };
BinPlan planBins(const std::vector<uint64_t> &lengths, uint64_t binBases)
{
    BinPlan plan;
    const uint64_t GRAIN = 2048;                                  // MatchDistribution::getBinSize: where the reference's bins can begin
    binBases = std::max<uint64_t>(GRAIN, (binBases + GRAIN - 1) / GRAIN * GRAIN);
    uint64_t open = 0;                                            // bases in the bin that is being filled with whole contigs
    for (uint32_t c = 0; c < lengths.size(); ++c)
    {
        const uint64_t length = lengths[c];
        if (length > binBases + binBases / 2)
        {   // a contig of several bins: equal stretches that begin on the grain
            const uint64_t pieces = (length + binBases - 1) / binBases;
            const uint64_t stretch = ((length + pieces - 1) / pieces + GRAIN - 1) / GRAIN * GRAIN;
            plan.binOfContig.push_back(uint32_t(plan.ranges.size()));
            for (uint64_t at = 0; at < length; at += stretch)
            {
                if (at) plan.cuts.push_back(referencePosition(c, at));
                plan.ranges.emplace_back(referencePosition(c, at), at + stretch < length ? referencePosition(c, at + stretch) : referencePosition(c + 1, 0));
            }
            open = 0;
            continue;
        }
        if (!open || open + length > binBases) { plan.ranges.emplace_back(referencePosition(c, 0), referencePosition(c + 1, 0)); open = 0; }
        else plan.ranges.back().second = referencePosition(c + 1, 0);
        plan.binOfContig.push_back(uint32_t(plan.ranges.size() - 1));
        open += std::max<uint64_t>(length, 1);
    }
    return plan;
}

// a device and what runs on it
struct Worker
{
    int device = 0; unsigned id = 0;
    int place = 0;                                  // which device's memory the worker's blocks are in, as the host sees it: the device, unless a test makes the workers of one device strangers
    isaac_gpu_ctx *ctx = 0;
    std::mutex ctxLock;                             // load phase: the lanes' threads take turns on the context
    std::vector<Tile *> tiles;
    DeviceMemory matches, offsets, textDev;
    uint64_t matchCapacity = 0;
    std::vector<uint8_t> contigHasMatches;
    isaac_counters counters;
    uint64_t tilesKeptOnDevice = 0, loadsKeptOnDevice = 0, peakDeviceBytes = 0, mapqResolved = 0, mapqChanged = 0;
    double selectSeconds = 0, buildSeconds = 0, uploadSeconds = 0, recordsSeconds = 0, deflateSeconds = 0, downloadSeconds = 0, writerWaitSeconds = 0, releaseSeconds = 0;
    ~Worker() { matches.release(); offsets.release(); textDev.release(); if (ctx) isaac_gpu_destroy(ctx); }
    void noteMemory() { uint64_t f = 0, t = 0; if (!isaac_gpu_memory_info(ctx, &f, &t)) peakDeviceBytes = std::max(peakDeviceBytes, t - f); }
};

// Page-locked host buffers handed round between the builders (which fill them from the device at the link's rate: into pageable memory the same copy runs
// at a third of it) and the writer (which gives them back): pinning memory is slow, so the buffers are kept and grow to the largest request
const size_t BUILD_AHEAD = std::getenv("ISAAC_ALIGN_BUILD_AHEAD") ? size_t(std::max(1, std::atoi(std::getenv("ISAAC_ALIGN_BUILD_AHEAD")))) : 8;            // finished bins that may wait in host memory for the writer (the variable: measurements)
class PinnedPool
{
public:
    struct Buffer { void *p = 0; uint64_t bytes = 0; };
    ~PinnedPool() { for (Buffer &b : free_) if (b.p) isaac_gpu_host_free(b.p); }
    Buffer take(uint64_t bytes)
    {
        Buffer best;
        {
            std::lock_guard<std::mutex> hold(lock_);
            size_t at = free_.size();
            for (size_t i = 0; i < free_.size(); ++i) if (free_[i].bytes >= bytes && (at == free_.size() || free_[i].bytes < free_[at].bytes)) at = i;
            if (at == free_.size() && !free_.empty()) { at = 0; for (size_t i = 1; i < free_.size(); ++i) if (free_[i].bytes > free_[at].bytes) at = i; }      // the largest one is replaced
            if (at < free_.size()) { best = free_[at]; free_.erase(free_.begin() + long(at)); }
        }
        if (best.bytes < bytes)
        {
            if (best.p) isaac_gpu_host_free(best.p);
            best.bytes = bytes + bytes / 8 + 4096;
            GPU(isaac_gpu_host_malloc(best.bytes, &best.p));
        }
        return best;
    }
    void give(Buffer b) { if (b.p) { std::lock_guard<std::mutex> hold(lock_); free_.push_back(b); } }
private:
    std::mutex lock_; std::vector<Buffer> free_;
};
// the matches of a tile and where they begin per cluster, grown as isaac_gpu_find_matches asks
struct Finder { DeviceMemory matches, offsets; uint64_t capacity = 0; };
// a context per device for the lanes' threads (text upload, conversion, first lookup), beside the workers' own: loading does not wait for a selection
struct Loader
{
    int place = 0; unsigned reader = 0, read = 0; isaac_gpu_ctx *ctx = 0; std::mutex lock; DeviceMemory textDev; Finder finder; std::vector<uint8_t> hits;
    void close() { textDev.release(); finder.matches.release(); finder.offsets.release(); if (ctx) isaac_gpu_destroy(ctx); ctx = 0; }
    ~Loader() { close(); }
};

// what a bin of the file becomes: its BGZF blocks, and what the index wants to know about its records
struct BinOutput
{
    bool ready = false; PinnedPool::Buffer bgzf, entries; uint64_t bgzfBytes = 0, recordsBytes = 0, nRecords = 0; std::string error;
    // a bin of several small contigs is compressed contig by contig: bam::BamIndex takes a run of BGZF blocks per contig (the reference's bins never
    // span contigs, include/alignment/matchSelector/BinIndexMap.hh:78-83), and so does the index written here
    struct Segment { uint64_t bgzfOffset, bgzfBytes, recordsOffset, recordsBytes, firstEntry, nEntries; };
    std::vector<Segment> segments;
};

// `bytes` at `offset` of the file (its end so far).  A large piece is copied into a shared mapping of the file by several threads side by side: write()
// calls on one file take turns (the inode's lock), which held a 28 GB sorted.bam to 4 GB/s; page faults on a mapping do not.  Files that cannot be
// mapped, and small pieces, are written.
void writeAt(int fd, const uint8_t *data, uint64_t bytes, uint64_t offset)
{
    const auto plain = [&]()
    {
        uint64_t done = 0;
        while (done < bytes)
        {
            const ssize_t r = ::pwrite(fd, data + done, size_t(bytes - done), off_t(offset + done));
            if (r < 0) { if (EINTR == errno) continue; throw std::runtime_error(std::string("Failed to write the BAM file: ") + std::strerror(errno)); }
            done += uint64_t(r);
        }
    };
    // (measured on the MI355X host, tmpfs: one writer 6 GB/s, eight writers on one file 4.3 GB/s, sixteen threads through a mapping 2.8 GB/s: the plain write
    // is the default, ISAAC_ALIGN_MAPPED_WRITES=1 selects the mapping)
    if (bytes < (uint64_t(32) << 20) || !std::getenv("ISAAC_ALIGN_MAPPED_WRITES")) { plain(); return; }
    const uint64_t page = uint64_t(::sysconf(_SC_PAGESIZE)), mapFrom = offset / page * page, lead = offset - mapFrom;
    if (::ftruncate(fd, off_t(offset + bytes))) { plain(); return; }
    void *map = ::mmap(0, size_t(lead + bytes), PROT_READ | PROT_WRITE, MAP_SHARED, fd, off_t(mapFrom));
    if (MAP_FAILED == map) { plain(); return; }
    uint8_t *to = static_cast<uint8_t *>(map) + lead;
    const unsigned threads = 16;
    const uint64_t share = ((bytes + threads - 1) / threads + page - 1) / page * page;
    std::vector<std::thread> workers;
    const auto part = [&](unsigned t) { const uint64_t begin = std::min<uint64_t>(bytes, t * share), end = std::min<uint64_t>(bytes, begin + share); if (end > begin) std::memcpy(to + begin, data + begin, size_t(end - begin)); };
    for (unsigned t = 1; t < threads; ++t) workers.emplace_back(part, t);
    part(0);
    for (std::thread &w : workers) w.join();
    ::munmap(map, size_t(lead + bytes));
}

// The output file's blocks, asked for ahead of the data (posix fallocate beyond the end of the file, a quarter of a gigabyte a call so that writes get their turn at
// the inode): on a disk that reserves extents; on tmpfs -- where the timed runs write -- it takes the page allocation, half of what a write() costs there, out of
// the writer's way and into the stages before it.  What the estimate asked for beyond the file's end is given back when the file is closed (ftruncate).
class Preallocator
{
public:
    Preallocator(int fd, uint64_t bytes) : fd_(fd), bytes_(bytes), stop_(false)
    {
        if (fd_ < 0 || !bytes_ || std::getenv("ISAAC_ALIGN_NO_PREALLOCATION")) return;
        struct statvfs fs;
        if (::fstatvfs(fd_, &fs) || uint64_t(fs.f_bavail) * fs.f_frsize / 2 < bytes_) return;       // never more than half of what is free
        thread_ = std::thread([this]()
        {
            const uint64_t step = uint64_t(256) << 20;
            for (uint64_t at = 0; at < bytes_ && !stop_.load(); at += step)
                if (::fallocate(fd_, FALLOC_FL_KEEP_SIZE, off_t(at), off_t(std::min(step, bytes_ - at)))) break;        // (a file system without it: the writes allocate as they go)
                else done_.store(at + std::min(step, bytes_ - at));
        });
    }
    void finish() { stop_.store(true); if (thread_.joinable()) thread_.join(); }
    uint64_t done() const { return done_.load(); }
    ~Preallocator() { finish(); }
private:
    int fd_; uint64_t bytes_; std::atomic<bool> stop_; std::atomic<uint64_t> done_{0}; std::thread thread_;
};

uint64_t hostResidentBytes()
{   // VmHWM: the process's peak resident set
    std::ifstream status("/proc/self/status");
    std::string line;
    while (std::getline(status, line)) if (0 == line.compare(0, 6, "VmHWM:")) return uint64_t(std::strtoull(line.c_str() + 6, 0, 10)) * 1024;
    return 0;
}

int run(const AlignOptions &o)
{
    const double runStart = seconds();
    // ---- flowcells (AlignOptions.cpp:1178-1290)
    std::vector<FastqFlowcell> flowcells;
    for (size_t i = 0; i < o.baseCalls.size(); ++i)
    {
        FastqFlowcell fc = FastqFlowcell::discover(o.baseCalls[i], "fastq-gz" == o.baseCallsFormat[i], o.laneNumberMax, o.useBasesMask, o.variableReadLength || o.variableFastqReadLength);
        const std::string original = fc.flowcellId;
        for (unsigned conflicts = 1; flowcells.end() != std::find_if(flowcells.begin(), flowcells.end(), [&fc](const FastqFlowcell &other) { return other.flowcellId == fc.flowcellId; }); ++conflicts)
            fc.flowcellId = original + "-" + std::to_string(conflicts);
        if (original != fc.flowcellId) std::cerr << "WARNING: renamed flowcell id " << original << " into " << fc.flowcellId << " to avoid duplication" << std::endl;
        if (!flowcells.empty() && (fc.nReads != flowcells[0].nReads || fc.readLength[0] != flowcells[0].readLength[0] || fc.readLength[1] != flowcells[0].readLength[1]))
            throw InvalidOption("\n   *** flowcells with different read lengths in one run are not supported by this host ***\n");
        std::cerr << "isaac-align: flowcell " << fc.flowcellId << " in " << fc.baseCallsDirectory << ": " << fc.lanes.size() << " lane(s), " << fc.nReads << " read(s) of " << fc.readLength[0]
                  << (2 == fc.nReads ? "+" + std::to_string(fc.readLength[1]) : std::string()) << " cycles" << std::endl;
        flowcells.push_back(fc);
    }
    const unsigned nReads = flowcells[0].nReads, clusterLength = flowcells[0].readLength[0] + flowcells[0].readLength[1];
    const isaac_params params = o.params(flowcells[0].readLength[0], 2 == nReads ? flowcells[0].readLength[1] : 0);

    std::deque<Loader> loaders;                       // (declared before the workers and the lanes: destroyed after them)
    std::vector<std::unique_ptr<Worker> > workers;
    struct Lane
    {
        const FastqFlowcell *flowcell; const FastqLane *lane; unsigned ordinal = 0; std::string readGroup; std::deque<Tile> tiles; std::deque<Load> loads; std::string error;
        // barcodeTemplateLengthStatistics (one 'none' barcode per lane): learnt tile by tile, in tile order, until a tile gives stable ones (MatchSelector.cpp:395-412)
        isaac_tls tls; bool tlsStable = false, learning = false; unsigned tlsNext = 0;
    };
    std::deque<Lane> lanes;
    for (const FastqFlowcell &fc : flowcells)
        for (const FastqLane &lane : fc.lanes)
        {
            lanes.emplace_back();
            Lane &L = lanes.back();
            L.flowcell = &fc; L.lane = &lane; L.ordinal = unsigned(lanes.size() - 1); L.readGroup = std::to_string(lanes.size() - 1);   // one 'none' barcode per lane, numbered in the order of the lanes
            std::memset(&L.tls, 0, sizeof(L.tls));
        }
    if (lanes.size() > 4096) throw std::runtime_error("more than 4096 lanes");
    // the threads that read lanes: one more than there are workers (a lane's text comes from a file at the rate of one file)
    const unsigned nReaders = unsigned(std::max<size_t>(1, std::min(lanes.size(), o.deviceList().size() + 1)));
    // How many clusters the run will have, before it has been read: the size of every lane's first file over the length of its first record (compressed
    // files: taken to hold four times their size).  Only the bins' sizes depend on it; any plan gives a valid file, and the estimate -- unlike the count,
    // which is known when the last lane is read -- is there when the first tile wants its bins.
    uint64_t estimatedClusters = 0;
    for (Lane &L : lanes)
    {
        const std::string &path = L.lane->readPath[0];
        struct stat st;
        if (path.empty() || ::stat(path.c_str(), &st)) continue;
        size_t lines = 0, recordBytes = 0;
        if (S_ISREG(st.st_mode))
        {   // (only a regular file can be looked into twice: bytes peeked from a pipe are lost to the reader proper)
            FastqFileReader peek(path, L.flowcell->compressed);
            std::vector<char> head(1 << 16);
            const size_t got = peek.readInto(head.data(), head.size());
            for (size_t i = 0; i < got && lines < 4; ++i) { ++recordBytes; if ('\n' == head[i]) ++lines; }
        }
        if (lines < 4 || !recordBytes) recordBytes = 2 * size_t(L.flowcell->fileReadLength[0]) + 64;
        estimatedClusters += uint64_t(st.st_size) * (L.flowcell->compressed ? 4 : 1) / recordBytes;
    }
    const uint64_t binRecords = o.binRecords ? o.binRecords : DEFAULT_BIN_RECORDS;
    if (std::getenv("ISAAC_ALIGN_PLAN_ONLY"))
    {   // What the run would do, decided before a device is touched -- the threads that read lanes, the loads, whether the selection is streamed, the bins -- as one
        // line for the tests of the planning (no HIP device is needed for it: sorted-reference.xml and the sizes of the FASTQ files are all it reads).
        const Reference ref = parseReference(o.referenceGenome);
        std::vector<uint64_t> lengths;
        for (const isaac_reference_contig &c : ref.contigs) lengths.push_back(c.total_bases);
        const double perBase = double(std::max<uint64_t>(estimatedClusters, 1)) * nReads / double(std::max<uint64_t>(1, ref.totalBases));
        const BinPlan plan = planBins(lengths, uint64_t(std::min(1e15, double(binRecords) / std::max(perBase, 1e-9))));
        const uint32_t tileMax = isaac_gpu_fastq_tile_clusters_max(o.clustersAtATime, params.n_seeds);
        const uint32_t load = o.clustersAtATime ? o.clustersAtATime : 4 * tileMax;
        const uint64_t loads = (estimatedClusters + load - 1) / std::max<uint64_t>(1, load);
        const char *streamSwitch = std::getenv("ISAAC_ALIGN_STREAM_SELECTION");
        const size_t nWorkers = o.deviceList().size();
        std::string ranges = "[";
        for (size_t b = 0; b < plan.ranges.size(); ++b) ranges += (b ? ", [" : "[") + std::to_string(plan.ranges[b].first) + ", " + std::to_string(plan.ranges[b].second) + "]";
        std::cout << "{\"estimated_clusters\": " << estimatedClusters << ", \"lanes\": " << lanes.size() << ", \"readers\": " << nReaders << ", \"workers\": " << nWorkers
                  << ", \"loader_contexts\": " << nReaders * nReads << ", \"tile_clusters_max\": " << tileMax << ", \"load_clusters\": " << load << ", \"expected_loads\": " << loads
                  << ", \"selection_streamed\": " << ((streamSwitch ? 0 != std::atoi(streamSwitch) : loads >= 8 * nWorkers) ? 1 : 0)
                  << ", \"bins\": " << plan.ranges.size() + 1 << ", \"bin_cuts\": " << plan.cuts.size() << ", \"bin_ranges\": " << ranges << "]}" << std::endl;
        return 0;
    }
    // the page-locked buffers the build stage fills are made while the reference is read and its table copied (locking pages is slow -- a gigabyte takes a tenth of a second and more -- and holds other calls of the runtime up: beside the base calls' loading it cost that stage as much as it saved the later one)
    PinnedPool pinned;
    std::thread pinnedWarm([&]()
    {
        try
        {
            const uint64_t perBin = std::min<uint64_t>(binRecords, std::max<uint64_t>(estimatedClusters * nReads, 1));
            std::vector<PinnedPool::Buffer> made;
            // as many as the build stage can have in use: the bins that wait for the writer (BUILD_AHEAD), and per worker the one in work and the one whose blocks are on their way
            const uint64_t expectedBins = (std::max<uint64_t>(estimatedClusters * nReads, 1) + binRecords - 1) / binRecords + 1;
            const unsigned buffers = std::getenv("ISAAC_ALIGN_WARM_BUFFERS") ? unsigned(std::atoi(std::getenv("ISAAC_ALIGN_WARM_BUFFERS"))) : unsigned(std::min<uint64_t>(expectedBins, BUILD_AHEAD + 2 * o.deviceList().size()));
            for (unsigned i = 0; i < buffers; ++i) { made.push_back(pinned.take(perBin * 160)); made.push_back(pinned.take(perBin * sizeof(isaac_bam_index_entry))); }
            for (PinnedPool::Buffer &b : made) pinned.give(b);
        }
        catch (const std::exception &) {}       // (the build stage asks again and reports what fails)
    });
    struct JoinWarm { std::thread &t; ~JoinWarm() { if (t.joinable()) t.join(); } } joinWarm{ pinnedWarm };
    // ---- the output file, opened now: its blocks are asked for while everything before the first write runs
    const std::string directory = o.outputDirectory + "/Projects/default/default";
    makeDirectories(directory);
    const std::string bamPath = directory + "/sorted.bam";
    struct OutputFile { int fd = -1; ~OutputFile() { if (fd >= 0) ::close(fd); } } outputFile;
    outputFile.fd = ::open(bamPath.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (outputFile.fd < 0) throw std::runtime_error("Failed to open output BAM file " + bamPath);
    // (a record is some 110 bytes and three halves of its bases; deflated -- base qualities are most of it -- about half of that)
    const uint64_t expectedBamBytes = uint64_t(double(estimatedClusters) * nReads * (110.0 + 1.5 * clusterLength / nReads) * (o.bamGzipLevel ? 0.55 : 1.02));
    Preallocator preallocator(outputFile.fd, expectedBamBytes);

    // ---- the workers and the reference: the first context of a device loads the table, the others of that device share it; another device
    // gets a copy over the link between the two (isaac_gpu_copy)
    const std::vector<int> devices = o.deviceList();
    // ISAAC_ALIGN_STRANGERS (tests on a box with one device): the workers of one device treat each other's memory as another device's -- the table is
    // copied, a bin's parts on the other worker's blocks are fetched -- which is every line two devices run
    const bool strangers = 0 != std::getenv("ISAAC_ALIGN_STRANGERS");
    Reference reference;
    double fastaSeconds = 0, contigSeconds = 0, tableSeconds = 0, shareSeconds = 0;
    {
        Stage stage("loading the reference");
        reference = loadReference(o.referenceGenome);
        fastaSeconds = seconds() - runStart;
        for (size_t k = 0; k < devices.size(); ++k)
        {
            workers.emplace_back(new Worker);
            Worker &w = *workers.back();
            w.device = devices[k]; w.id = unsigned(k); w.place = strangers ? int(1000 + k) : devices[k];
            GPU(isaac_gpu_create(w.device, &params, ISAAC_GPU_STREAM_OWN, &w.ctx));
            if (0 == k)
            {
                const double contigStart = seconds();
                GPU(isaac_gpu_load_contigs(w.ctx, reference.bases.get(), reference.offsets.data(), uint32_t(reference.contigs.size())));
                contigSeconds += seconds() - contigStart;
                // the bases stay in host memory: the few clusters per million whose MAPQ arithmetic wants glibc's word (isaac_gpu_resolve_flagged) read them from here
                GPU(isaac_gpu_set_host_contigs(w.ctx, reference.bases.get()));
                const double tableStart = seconds();
                GPU(isaac_gpu_load_sorted_reference(w.ctx, o.referenceGenome.c_str()));
                tableSeconds = seconds() - tableStart;
                continue;
            }
            // everybody else reads what is there: a worker of a device that has one takes that one's contigs and table in place, the first worker of another
            // device gets copies over the link between the two
            const double shareStart = seconds();
            Worker *samePlace = 0;
            for (size_t j = 0; j < k && !samePlace; ++j) if (workers[j]->place == w.place) samePlace = workers[j].get();
            GPU(isaac_gpu_share_reference(w.ctx, samePlace ? samePlace->ctx : workers[0]->ctx));
            shareSeconds += seconds() - shareStart;
        }
        // the loaders: per device a context for every thread that reads lanes and every read of a lane (a context costs a stream and its scratch, the reference
        // is the worker's) -- one read's text is on the link while the other's is converted, and so are two lanes'
        for (auto &w : workers)
        {
            bool have = false;
            for (Loader &l : loaders) if (l.place == w->place) have = true;
            if (have) continue;
            const double shareStart = seconds();
            for (unsigned reader = 0; reader < nReaders; ++reader)
                for (unsigned read = 0; read < nReads; ++read)
                {
                    loaders.emplace_back();
                    Loader &l = loaders.back();
                    l.place = w->place; l.reader = reader; l.read = read; l.hits.assign(reference.contigs.size(), 0);
                    GPU(isaac_gpu_create(w->device, &params, ISAAC_GPU_STREAM_OWN, &l.ctx));
                    GPU(isaac_gpu_share_reference(l.ctx, w->ctx));
                }
            shareSeconds += seconds() - shareStart;
        }
    }
    const uint32_t nContigs = uint32_t(reference.contigs.size());
    const double referenceSeconds = seconds() - runStart;

    // ---- FastqSeedSource + FindMatchesTransition + SelectMatchesTransition as one pipeline.  The lanes are read side by side (a thread each): loads of
    // --clusters-at-a-time clusters, tiles of at most tileClustersMax; every load of a lane is dealt to the next worker in turn and looked up once for the
    // contigs its matches touch.  The reference finishes that pass over the whole run before it selects anything, because the selection wants to know which
    // contigs have matches anywhere (MatchSelector loads only those; the rest-of-genome correction counts them).  Once every contig has a match no later
    // lookup can change that set: from then on a tile is selected -- on its worker's context, by the worker's own thread -- as soon as it is loaded, while
    // the lanes' threads go on reading and converting on a context of their own (one per device: the loader).  A run that never gets there (a small
    // reference with untouched contigs) selects after the last load, as the reference does.
    const uint32_t tileClustersMax = isaac_gpu_fastq_tile_clusters_max(o.clustersAtATime, params.n_seeds);
    const uint32_t loadClusters = o.clustersAtATime ? o.clustersAtATime : 4 * tileClustersMax;      // a multiple of the tile size: the tiles come out the same for any such load
    auto findMatches = [&](isaac_gpu_ctx *ctx, Finder &f, const uint8_t *bcl, uint32_t clusters, uint32_t tileIndex, uint64_t &nMatches, uint8_t *hits)
    {
        const uint64_t worst = uint64_t(clusters) * 2 * params.n_seeds * std::max(1u, params.repeat_threshold - 1);
        if (!f.offsets.bytes()) f.offsets.reset(ctx, (uint64_t(tileClustersMax) + 1) * 8);
        if (!f.capacity) { f.capacity = std::max<uint64_t>(1024, std::min<uint64_t>(worst, uint64_t(tileClustersMax) * 24)); f.matches.reset(ctx, f.capacity * sizeof(isaac_match)); }
        for (;;)
        {
            const int rc = isaac_gpu_find_matches(ctx, bcl, clusters, tileIndex & 0xfffu /* SeedId's tile field; nothing reads it back */, f.matches.as<isaac_match>(), f.capacity, f.offsets.as<uint64_t>(), &nMatches, hits);
            if (ISAAC_GPU_ECAPACITY != rc) { check(rc, "isaac_gpu_find_matches"); return; }
            f.capacity = std::max(nMatches, 2 * f.capacity);
            f.matches.reset(ctx, f.capacity * sizeof(isaac_match));
        }
    };
    auto loaderOf = [&](const Worker &w, unsigned reader, unsigned read) -> Loader &
    {
        for (Loader &l : loaders) if (l.place == w.place && l.reader == reader && l.read == read) return l;
        throw std::logic_error("no loader");
    };
    // the BCL bytes of a load stay on the device while it keeps this much free for the selection's scratch and the bins (ISAAC_ALIGN_HOST_LOADS: tests)
    const bool hostLoads = 0 != std::getenv("ISAAC_ALIGN_HOST_LOADS");
    const bool skipResolution = 0 != std::getenv("ISAAC_ALIGN_TIMING_NO_RESOLUTION");       // (timing only: what isaac_gpu_resolve_flagged costs)
    auto keepOnDevice = [&](isaac_gpu_ctx *ctx, uint64_t bytes)
    {
        if (hostLoads) return false;
        uint64_t freeBytes = 0, totalBytes = 0;
        GPU(isaac_gpu_memory_info(ctx, &freeBytes, &totalBytes));
        return freeBytes > totalBytes * 2 / 5 + bytes;
    };
    // io::FastqLoader::loadSingleRead for up to maxClusters clusters: the text goes to the device in pieces, the converter leaves the incomplete
    // record at the end of a piece for the next one
    auto loadRead = [&](Loader &l, TextStream &stream, unsigned readIndex, uint8_t *bclDev, uint32_t maxClusters) -> uint32_t
    {
        const bool allowVariableLength = o.variableReadLength || o.variableFastqReadLength;
        uint32_t clusters = 0;
        while (clusters < maxClusters)
        {
            if (stream.size() < TEXT_SLACK && !stream.final()) stream.advance();
            if (!stream.size()) break;
            const bool final = stream.final();
            const double convertStart = seconds();
            uint32_t n = 0; uint64_t consumed = 0, errorOffset = 0;
            {
                std::lock_guard<std::mutex> turn(l.lock);
                if (l.textDev.bytes() < stream.size() + 64) l.textDev.reset(l.ctx, TEXT_SLACK + TEXT_CHUNK + 64);
                GPU(isaac_gpu_upload(l.ctx, l.textDev.as<char>(), stream.data(), stream.size()));
                const int rc = isaac_gpu_fastq_to_bcl(l.ctx, l.textDev.as<char>(), stream.size(), readIndex, allowVariableLength, final, bclDev + uint64_t(clusters) * clusterLength,
                                                      maxClusters - clusters, &n, &consumed, &errorOffset);
                if (rc)
                    throw std::runtime_error(stream.path() + ": " + isaac_gpu_last_error() + " (record " + std::to_string(clusters + n) + " of this load, offset " +
                                             std::to_string(stream.consumedBytes + errorOffset) + ")");
            }
            addTime(g_convertSeconds, seconds() - convertStart);
            clusters += n;
            stream.consume(consumed);
            if (!n && !consumed)
            {
                if (final) break;                                     // nothing but line ends left
                if (stream.size() >= TEXT_SLACK) throw std::runtime_error(stream.path() + ": a record of more than " + std::to_string(TEXT_SLACK) + " bytes");
            }
        }
        return clusters;
    };
    // ---- the bins (see BinPart): sized for --bin-records records each, by the reads the run is expected to have per base of the reference
    std::vector<uint64_t> contigLengths;
    for (const isaac_reference_contig &c : reference.contigs) contigLengths.push_back(c.total_bases);
    const double recordsPerBase = double(std::max<uint64_t>(estimatedClusters, 1)) * nReads / double(std::max<uint64_t>(1, reference.totalBases));
    const BinPlan plan = planBins(contigLengths, uint64_t(std::min(1e15, double(binRecords) / std::max(recordsPerBase, 1e-9))));
    const uint32_t nBins = uint32_t(plan.ranges.size()) + 1;
    if (nBins > 65535) throw std::runtime_error("more than 65534 bins: a larger --bin-records is needed");
    std::vector<Bin> bins(nBins);
    for (uint32_t b = 0; b + 1 < nBins; ++b) { bins[b].firstPosition = plan.ranges[b].first; bins[b].endPosition = plan.ranges[b].second; }
    bins[nBins - 1].unaligned = true;
    isaac_bin_map binMap; binMap.bin_of_contig = plan.binOfContig.data(); binMap.n_contigs = nContigs; binMap.cut_positions = plan.cuts.data(); binMap.n_cuts = uint32_t(plan.cuts.size()); binMap.n_bins = nBins;
    std::cerr << "isaac-align: about " << estimatedClusters << " clusters expected: " << nBins - 1 << " bin(s) of about " << binRecords << " records for " << nContigs << " contig(s), " << plan.cuts.size()
              << " cut(s) inside contigs" << std::endl;

    // ---- what the threads share
    std::mutex shared; std::condition_variable wake;
    std::vector<uint8_t> contigHasMatches(nContigs, 0);
    bool hitsClosed = false, loadingDone = false; std::string pipelineError;
    std::vector<std::deque<Tile *> > ready(workers.size());          // per worker: its tiles that are loaded and not yet selected, in the order they were loaded
    std::vector<uint64_t> unselected(workers.size(), 0);             // tiles dealt to the worker and not yet through its selection
    std::atomic<uint64_t> totalClusters(0), totalTiles(0);
    const bool hostBins = 0 != std::getenv("ISAAC_ALIGN_HOST_BINS");          // tests: every part through host memory
    // -m / --memory-limit: gigabytes the parts in host memory may take; what comes after that goes to the bins' files under -t (ISAAC_ALIGN_SPILL_BINS: tests, every such part)
    const bool spillBins = 0 != std::getenv("ISAAC_ALIGN_SPILL_BINS");
    const uint64_t hostPartLimit = uint64_t(o.memoryLimit) << 30;
    std::atomic<uint64_t> hostPartBytes(0), spilledBytes(0);
    // The selection may begin as soon as every contig has a match (closeHits).  It does when the run is expected to be long (eight loads per worker and more): the selection
    // of the early loads then runs beside the reading and conversion of the later ones.  Both are bound by the device, so what streaming hides is the shorter of the
    // two stages, and only where there is enough of a run to hide it in: 100 M pairs in 28 loads on two workers gain between nothing and a second of fourteen
    // (profiles/r5_cli_headline_100M_*.json); 10 M pairs in five loads gained 0.15 s stand-alone and lost 0.3 s as bench.py's leg (profiles/r5_r_cli_timing.log,
    // r5_last_bench_default.json against r5_j), where the stages only get in each other's way.  ISAAC_ALIGN_STREAM_SELECTION=1 / 0 forces either.
    const uint64_t expectedLoads = (estimatedClusters + loadClusters - 1) / std::max<uint64_t>(1, loadClusters);
    const char *streamSwitch = std::getenv("ISAAC_ALIGN_STREAM_SELECTION");
    const bool streamSelection = streamSwitch ? 0 != std::atoi(streamSwitch) : expectedLoads >= 8 * workers.size();
    // ISAAC_ALIGN_DUMP_TILES=<directory>:<lane>.<tile>,...: tiles (by lane number and tile number, as in the read names) to write out as they were selected
    std::set<std::pair<unsigned, unsigned> > dumpTiles; std::string dumpDirectory;
    if (const char *e = std::getenv("ISAAC_ALIGN_DUMP_TILES"))
    {
        const std::string spec(e);
        const size_t colon = spec.rfind(':');
        if (std::string::npos == colon) throw std::runtime_error("ISAAC_ALIGN_DUMP_TILES=<directory>:<lane>.<tile>,...");
        dumpDirectory = spec.substr(0, colon);
        std::stringstream list(spec.substr(colon + 1));
        for (std::string item; std::getline(list, item, ','); )
        {
            const size_t dot = item.find('.');
            if (item.empty() || std::string::npos == dot) continue;
            dumpTiles.insert(std::make_pair(unsigned(std::stoul(item.substr(0, dot))), unsigned(std::stoul(item.substr(dot + 1)))));
        }
        makeDirectories(dumpDirectory);
    }
    // every contig has a match (or the last load is in): the set is final, the workers' contexts learn it, the selection may begin.  Called with `shared` held.
    auto closeHits = [&]()
    {
        for (auto &w : workers) GPU(isaac_gpu_set_loaded_contigs(w->ctx, contigHasMatches.data(), nContigs));       // (the workers are idle until now)
        hitsClosed = true;
        wake.notify_all();
    };
    // a load's BCL bytes on its worker's device (where they may have been all along)
    auto loadOnDevice = [&](Load &load) -> const uint8_t *
    {
        if (!load.dev.bytes())
        {
            load.dev.reset(load.worker->ctx, load.bytes + 64);
            GPU(isaac_gpu_upload(load.worker->ctx, load.dev.as<uint8_t>(), load.host.get(), load.bytes));
        }
        return load.dev.as<uint8_t>();
    };

    // ---- the lanes' threads
    const double loadStart = seconds();
    std::atomic<size_t> nextLane(0), nextWorker(0);
    auto readLanes = [&](unsigned reader)
    {
        for (size_t k = nextLane++; k < lanes.size(); k = nextLane++)
        {
            Lane &L = lanes[k];
            try
            {
                const FastqFlowcell &fc = *L.flowcell; const FastqLane &lane = *L.lane;
                std::unique_ptr<TextStream> streams[2];
                struct OpenTime { double start; ~OpenTime() { addTime(g_textOpenSeconds, seconds() - start); } };
                for (unsigned r = 0; r < nReads; ++r)
                {
                    OpenTime timed{ seconds() };
                    if (lane.readPath[r].empty()) throw std::runtime_error("lane " + std::to_string(lane.lane) + " of " + fc.baseCallsDirectory + " has no read " + std::to_string(r + 1));
                    streams[r].reset(new TextStream(lane.readPath[r], fc.compressed));
                }
                uint32_t nextTile = 1;
                for (;;)
                {
                    { std::lock_guard<std::mutex> hold(shared); if (!pipelineError.empty()) break; }
                    Worker &w = *workers[nextWorker++ % workers.size()];
                    Loader &l = loaderOf(w, reader, 0);
                    DeviceMemory bcl;
                    { const double start = seconds(); std::lock_guard<std::mutex> turn(l.lock); bcl.reset(l.ctx, uint64_t(loadClusters) * clusterLength + 64); addTime(g_loadMemorySeconds, seconds() - start); }
                    uint32_t loaded[2] = { 0, 0 };
                    {   // the second read beside the first, on a context of its own: the bytes of a cluster's two reads do not overlap
                        std::string secondError;
                        std::thread second;
                        if (2 == nReads)
                            second = std::thread([&]()
                            {
                                try { loaded[1] = loadRead(loaderOf(w, reader, 1), *streams[1], 1, bcl.as<uint8_t>(), loadClusters); }
                                catch (const std::exception &e) { secondError = e.what(); }
                            });
                        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{ second };
                        loaded[0] = loadRead(l, *streams[0], 0, bcl.as<uint8_t>(), loadClusters);
                        if (second.joinable()) second.join();
                        if (!secondError.empty()) throw std::runtime_error(secondError);
                    }
                    if (2 == nReads && loaded[0] != loaded[1])
                        throw std::runtime_error("Mismatching number of clusters in " + lane.readPath[0] + " (" + std::to_string(loaded[0]) + ") and " + lane.readPath[1] + " (" + std::to_string(loaded[1]) + ")");
                    if (!loaded[0]) break;
                    const uint64_t bytes = uint64_t(loaded[0]) * clusterLength;
                    uint32_t nTiles = 0, next = 0;
                    isaac_gpu_fastq_tiles(loaded[0], o.clustersAtATime, params.n_seeds, nextTile, 0, 0, 0, &nTiles, &next);
                    std::vector<uint32_t> numbers(nTiles), sizes(nTiles);
                    GPU(isaac_gpu_fastq_tiles(loaded[0], o.clustersAtATime, params.n_seeds, nextTile, numbers.data(), sizes.data(), nTiles, &nTiles, &next));
                    nextTile = next;
                    L.loads.emplace_back();
                    Load &load = L.loads.back();
                    load.worker = &w; load.clusters = loaded[0]; load.bytes = bytes;
                    {
                        std::lock_guard<std::mutex> turn(l.lock);
                        uint64_t first = 0;
                        for (uint32_t i = 0; i < nTiles; ++i)
                        {
                            L.tiles.emplace_back();
                            Tile &t = L.tiles.back();
                            // the tile's index over the run (FragmentHeader::tile_): lanes in their order, tiles in the order of the lane's files -- known at once, which a
                            // running number over the run would not be while the lanes are read side by side.  Only the order of the indexes matters to the output.
                            t.lane = lane.lane; t.number = numbers[i]; t.index = (L.ordinal << 16) | unsigned(L.tiles.size() - 1); t.clusters = sizes[i]; t.load = &load; t.firstCluster = first; t.worker = &w;
                            if (L.tiles.size() > 65535) throw std::runtime_error("more than 65535 tiles in a lane");
                            t.namePrefix = fc.flowcellId + ":" + std::to_string(lane.lane) + ":" + std::to_string(t.number) + ":"; t.readGroup = L.readGroup;
                            std::memset(&t.tls, 0, sizeof(t.tls));
                            // the lookup that says which contigs have matches (the matches themselves are found again when the tile is selected)
                            uint64_t nMatches = 0;
                            const double lookupStart = seconds();
                            findMatches(l.ctx, l.finder, bcl.as<uint8_t>() + first * clusterLength, t.clusters, 0, nMatches, l.hits.data());
                            addTime(g_firstLookupSeconds, seconds() - lookupStart);
                            first += sizes[i];
                            load.tiles.push_back(&t);
                        }
                        load.tilesLeft = unsigned(load.tiles.size());
                        // where the load waits for the selection: the device while it has room, else host memory; and only the bytes it has
                        const double placeStart = seconds();
                        struct PlaceTime { double start; ~PlaceTime() { addTime(g_loadPlaceSeconds, seconds() - start); } } placeTime{ placeStart };
                        if (keepOnDevice(l.ctx, bytes))
                        {
                            if (loaded[0] < loadClusters / 2)
                            {   // "allocated too much memory for bcl data": the load keeps what it uses
                                DeviceMemory exact(l.ctx, bytes + 64);
                                GPU(isaac_gpu_copy(l.ctx, exact.as<uint8_t>(), bcl.as<uint8_t>(), bytes));
                                GPU(isaac_gpu_synchronize(l.ctx));
                                bcl = std::move(exact);
                            }
                            load.dev = std::move(bcl);
                            ++w.loadsKeptOnDevice;
                        }
                        else
                        {
                            load.host.reset(new uint8_t[bytes]);
                            GPU(isaac_gpu_download(l.ctx, load.host.get(), bcl.as<uint8_t>(), bytes));
                            bcl.release();
                        }
                    }
                    totalClusters += loaded[0]; totalTiles += nTiles;
                    {   // the load's tiles are its worker's to select; the contigs seen so far may complete the set
                        std::lock_guard<std::mutex> hold(shared);
                        for (Tile *t : load.tiles) { ready[w.id].push_back(t); ++unselected[w.id]; }
                        bool all = true;
                        for (uint32_t c = 0; c < nContigs; ++c) { contigHasMatches[c] |= l.hits[c]; all = all && contigHasMatches[c]; }
                        if (all && !hitsClosed && streamSelection) closeHits();
                        wake.notify_all();
                    }
                    if (loaded[0] < loadClusters) break;
                }
            }
            catch (const std::exception &e) { L.error = e.what(); std::lock_guard<std::mutex> hold(shared); if (pipelineError.empty()) pipelineError = e.what(); wake.notify_all(); }
        }
    };

    // ---- the workers' threads: SelectMatchesTransition for the worker's tiles, then BinningFragmentStorage
    auto selectTiles = [&](Worker &w)
    {
        try
        {
            Finder finder;
            DeviceMemory slots(w.ctx, uint64_t(tileClustersMax) * nReads * ISAAC_GPU_MAX_CIGAR_OPS * 4), records(w.ctx, uint64_t(tileClustersMax) * nReads * sizeof(isaac_fragment)), packed, binned;
            std::vector<isaac_bin_size> sizes(nBins);
            for (;;)
            {
                // the next tile of this worker that may go: any tile of a lane whose statistics are settled, or the tile a lane learns its statistics from next
                Tile *tp = 0; Lane *lane = 0; bool learns = false;
                {
                    std::unique_lock<std::mutex> hold(shared);
                    for (;;)
                    {
                        if (!pipelineError.empty()) return;
                        if (hitsClosed)
                        {
                            for (auto it = ready[w.id].begin(); it != ready[w.id].end() && !tp; ++it)
                            {
                                Lane &L = lanes[(*it)->index >> 16];
                                const unsigned ordinal = (*it)->index & 0xffffu;
                                const bool settled = L.tlsStable && !o.perTileTls;
                                if (settled || (ordinal == L.tlsNext && !L.learning))
                                {
                                    tp = *it; lane = &L; learns = !settled;
                                    if (learns) L.learning = true; else tp->tls = L.tls;
                                    ready[w.id].erase(it);
                                    break;
                                }
                            }
                            if (tp) break;
                            if (loadingDone && !unselected[w.id]) return;
                        }
                        wake.wait(hold);
                    }
                }
                Tile &t = *tp;
                const double start = seconds();
                const uint8_t *bcl = loadOnDevice(*t.load) + t.firstCluster * clusterLength;
                uint64_t nMatches = 0;
                findMatches(w.ctx, finder, bcl, t.clusters, t.index, nMatches, 0);
                if (learns)
                {
                    isaac_tls tls;
                    { std::lock_guard<std::mutex> hold(shared); tls = lane->tls; }
                    GPU(isaac_gpu_determine_tls(w.ctx, bcl, t.clusters, t.index, finder.matches.as<isaac_match>(), finder.offsets.as<uint64_t>(), &tls));
                    std::cerr << "isaac-align: template length statistics of tile " + t.namePrefix + " min " + std::to_string(tls.min) + " median " + std::to_string(tls.median) + " max " + std::to_string(tls.max) +
                                 (tls.stable ? " (stable)" : " (unstable)") + "\n";
                    t.tls = tls;
                    std::lock_guard<std::mutex> hold(shared);
                    lane->tls = tls; lane->tlsStable = 0 != tls.stable; lane->tlsNext = (t.index & 0xffffu) + 1; lane->learning = false;
                    wake.notify_all();
                }
                const uint64_t nRecords = uint64_t(t.clusters) * nReads;
                GPU(isaac_gpu_select_n(w.ctx, bcl, t.clusters, t.index, finder.matches.as<isaac_match>(), nMatches, finder.offsets.as<uint64_t>(), &t.tls, records.as<isaac_fragment>(), slots.as<uint32_t>(),
                                       nRecords * ISAAC_GPU_MAX_CIGAR_OPS));
                // the handful of clusters per million whose MAPQ arithmetic came within 1e-11 of an integer on the device take glibc's answer
                {
                    uint64_t flagged = 0, changed = 0;
                    const double resolveStart = seconds();
                    struct ResolveTime { double start; ~ResolveTime() { addTime(g_resolveSeconds, seconds() - start); } } resolveTime{ resolveStart };
                    if (!skipResolution)
                    GPU(isaac_gpu_resolve_flagged(w.ctx, bcl, t.clusters, t.index, finder.matches.as<isaac_match>(), finder.offsets.as<uint64_t>(), &t.tls, records.as<isaac_fragment>(), slots.as<uint32_t>(),
                                                  &flagged, &changed));
                    w.mapqResolved += flagged; w.mapqChanged += changed;
                }
                // the CIGARs as the bin files hold them: back to back
                uint64_t words = 0;
                if (packed.bytes() < nRecords * 8 * 4) packed.reset(w.ctx, nRecords * 8 * 4);
                int rc = isaac_gpu_compact_cigars(w.ctx, records.as<isaac_fragment>(), nRecords, slots.as<uint32_t>(), packed.as<uint32_t>(), packed.bytes() / 4, &words);
                if (ISAAC_GPU_ECAPACITY == rc)
                {
                    packed.reset(w.ctx, words * 4);
                    rc = isaac_gpu_compact_cigars(w.ctx, records.as<isaac_fragment>(), nRecords, slots.as<uint32_t>(), packed.as<uint32_t>(), packed.bytes() / 4, &words);
                }
                check(rc, "isaac_gpu_compact_cigars");
                if (dumpTiles.count(std::make_pair(t.lane, t.number)))
                {   // the tile as it was selected -- BCL bytes, records, packed CIGAR words -- for a checker
                    const std::string stem = dumpDirectory + "/tile_" + std::to_string(t.lane) + "_" + std::to_string(t.number);
                    const auto dump = [&](const std::string &path, const void *dev, uint64_t bytes)
                    {
                        std::vector<uint8_t> host(bytes);
                        if (bytes) GPU(isaac_gpu_download(w.ctx, host.data(), dev, bytes));
                        std::ofstream os(path.c_str(), std::ios::binary | std::ios::trunc);
                        if (!os.write(reinterpret_cast<const char *>(host.data()), std::streamsize(bytes))) throw std::runtime_error("Failed to write " + path);
                    };
                    dump(stem + ".bcl", bcl, uint64_t(t.clusters) * clusterLength);
                    dump(stem + ".records", records.as<isaac_fragment>(), nRecords * sizeof(isaac_fragment));
                    dump(stem + ".cigars", packed.as<uint32_t>(), words * 4);
                    std::ofstream meta((stem + ".json").c_str());
                    meta << "{\"index\": " << t.index << ", \"lane\": " << t.lane << ", \"number\": " << t.number << ", \"clusters\": " << t.clusters << ", \"read_group\": \"" << t.readGroup
                         << "\", \"tls\": [" << t.tls.min << ", " << t.tls.max << ", " << t.tls.median << ", " << t.tls.low_std_dev << ", " << t.tls.high_std_dev << ", " << t.tls.best_model[0] << ", "
                         << t.tls.best_model[1] << ", " << t.tls.stable << ", " << t.tls.mate_min << ", " << t.tls.mate_max << "]}" << std::endl;
                }
                // BinningFragmentStorage: the tile's clusters to their bins
                uint64_t need = 0;
                const uint64_t guess = align64(uint64_t(t.clusters) * clusterLength + nRecords * sizeof(isaac_fragment) + words * 4) * 5 / 4 + 256 * uint64_t(nBins);
                if (binned.bytes() < guess) binned.reset(w.ctx, guess);
                rc = isaac_gpu_bin_tile_map(w.ctx, bcl, records.as<isaac_fragment>(), packed.as<uint32_t>(), t.clusters, &binMap, binned.as<uint8_t>(), binned.bytes(), sizes.data(), &need);
                if (ISAAC_GPU_ECAPACITY == rc)
                {
                    binned.reset(w.ctx, need);
                    rc = isaac_gpu_bin_tile_map(w.ctx, bcl, records.as<isaac_fragment>(), packed.as<uint32_t>(), t.clusters, &binMap, binned.as<uint8_t>(), binned.bytes(), sizes.data(), &need);
                }
                check(rc, "isaac_gpu_bin_tile_map");
                // the BCL bytes of a load are in the bins once its last tile is: its memory is free for the loads that follow
                {
                    bool last;
                    { std::lock_guard<std::mutex> hold(shared); last = 0 == --t.load->tilesLeft; }
                    if (last) { t.load->dev.release(); t.load->host.reset(); }
                }
                // the tile's parts stay on the device while it has room (see BinPart); a block that is mostly slack is cut to size first
                std::shared_ptr<DeviceMemory> block;
                {
                    uint64_t freeBytes = 0, totalBytes = 0;
                    GPU(isaac_gpu_memory_info(w.ctx, &freeBytes, &totalBytes));
                    if (!hostBins && freeBytes > totalBytes / 4 + binned.bytes())
                    {
                        if (need + (need >> 2) < binned.bytes())
                        {
                            DeviceMemory exact(w.ctx, need + 64);
                            GPU(isaac_gpu_copy(w.ctx, exact.as<uint8_t>(), binned.as<uint8_t>(), need));
                            GPU(isaac_gpu_synchronize(w.ctx));
                            block = std::make_shared<DeviceMemory>(std::move(exact));
                        }
                        else block = std::make_shared<DeviceMemory>(std::move(binned));
                        ++w.tilesKeptOnDevice;
                    }
                }
                const uint8_t *binnedBytes = block ? block->as<uint8_t>() : binned.as<uint8_t>();
                std::unique_ptr<uint8_t[]> whole;
                if (!block && need) { whole.reset(new uint8_t[need]); GPU(isaac_gpu_download(w.ctx, whole.get(), binnedBytes, need)); }      // one transfer for the tile, cut up on the host
                uint64_t at = 0;
                for (uint32_t b = 0; b < nBins; ++b)
                {
                    const uint64_t m = sizes[b].n_clusters, cw = sizes[b].n_cigar_words;
                    const uint64_t bytes = align64(align64(m * clusterLength) + m * nReads * sizeof(isaac_fragment)) + align64(cw * 4);
                    if (m)
                    {
                        BinPart part; part.tile = &t; part.clusters = m; part.words = cw; part.bytes = bytes;
                        if (block) { part.block = block; part.offset = at; part.place = w.place; }
                        else if (spillBins || (hostPartLimit && hostPartBytes.load() + bytes > hostPartLimit)) part.spilled = true;
                        else { part.data.reset(new uint8_t[bytes]); std::memcpy(part.data.get(), whole.get() + at, bytes); hostPartBytes += bytes; }
                        std::lock_guard<std::mutex> guard(bins[b].lock);
                        if (part.spilled)
                        {
                            Bin &bin = bins[b];
                            if (bin.spillFd < 0)
                            {
                                makeDirectories(o.tempDirectory);
                                bin.spillPath = o.tempDirectory + "/isaac-align-bin-" + std::to_string(b) + "-" + std::to_string(::getpid()) + ".dat";
                                bin.spillFd = ::open(bin.spillPath.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
                                if (bin.spillFd < 0) throw std::runtime_error("Failed to create " + bin.spillPath + ": " + std::strerror(errno));
                            }
                            part.fileOffset = bin.spillBytes;
                            for (uint64_t done = 0; done < bytes; )
                            {
                                const ssize_t r = ::pwrite(bin.spillFd, whole.get() + at + done, size_t(bytes - done), off_t(part.fileOffset + done));
                                if (r < 0) { if (EINTR == errno) continue; throw std::runtime_error("Failed to write " + bin.spillPath + ": " + std::strerror(errno)); }
                                done += uint64_t(r);
                            }
                            bin.spillBytes += bytes; spilledBytes += bytes;
                        }
                        bins[b].bytes += bytes; bins[b].records += m * nReads;
                        bins[b].parts.push_back(std::move(part));
                    }
                    at += bytes;
                }
                w.noteMemory();
                w.selectSeconds += seconds() - start;
                { std::lock_guard<std::mutex> hold(shared); --unselected[w.id]; wake.notify_all(); }
            }
        }
        catch (const std::exception &e) { std::lock_guard<std::mutex> hold(shared); if (pipelineError.empty()) pipelineError = e.what(); wake.notify_all(); }
    };

    double loadSeconds = 0;
    const double selectStart = seconds();
    {
        Stage stage("loading base calls, finding and selecting matches");
        std::vector<std::thread> selectors, readers;
        for (auto &w : workers) selectors.emplace_back(selectTiles, std::ref(*w));
        for (unsigned k = 0; k < nReaders; ++k) readers.emplace_back(readLanes, k);
        for (std::thread &t : readers) t.join();
        loadSeconds = seconds() - loadStart;
        {
            std::lock_guard<std::mutex> hold(shared);
            loadingDone = true;
            if (!hitsClosed && pipelineError.empty())
            {
                try { closeHits(); } catch (const std::exception &e) { pipelineError = e.what(); }
            }
            wake.notify_all();
        }
        for (std::thread &t : selectors) t.join();
        for (const Lane &L : lanes) if (!L.error.empty()) throw std::runtime_error(L.error);
        if (!pipelineError.empty()) throw std::runtime_error(pipelineError);
        for (auto &w : workers) { GPU(isaac_gpu_synchronize(w->ctx)); isaac_gpu_get_counters(w->ctx, &w->counters); }
        // parts in tile order inside a bin, whichever worker was first
        for (Bin &bin : bins) std::sort(bin.parts.begin(), bin.parts.end(), [](const BinPart &a, const BinPart &b) { return a.tile->index < b.tile->index; });
    }
    // every load has been selected and its bytes released: the loaders' contexts -- contigs, directory, buffers -- make room for the build stage
    for (Lane &L : lanes) for (Load &load : L.loads) { load.dev.release(); load.host.reset(); }
    for (Loader &l : loaders) l.close();
    std::vector<Tile *> tiles;
    for (Lane &L : lanes) for (Tile &t : L.tiles) tiles.push_back(&t);
    std::cerr << "isaac-align: " << totalClusters.load() << " clusters in " << tiles.size() << " tile(s) on " << workers.size() << " worker(s)" << std::endl;
    if (tiles.empty()) throw InvalidOption("No data found to process. Please check your --base-calls.");
    uint64_t overflowClusters = 0, mapqNearInteger = 0, mapqResolved = 0, mapqChanged = 0;
    for (auto &w : workers) { overflowClusters += w->counters.overflow_clusters; mapqNearInteger += w->counters.mapq_near_integer; mapqResolved += w->mapqResolved; mapqChanged += w->mapqChanged; }
    const double selectSeconds = seconds() - selectStart - loadSeconds;       // what the selection took beyond the loading it ran beside

    // ---- build::Build: one bin at a time -- records, duplicates, realignment, BAM records, BGZF blocks on the device -- the file and its index
    // in bin order
    const double buildStart = seconds();
    Stage stage("building and writing sorted.bam");
    // header (Bam.hh:153-235): --bam-header-tag lines, the read groups in the order of a map keyed by their ids, the contigs in karyotype order
    std::vector<std::string> headerLines = o.bamHeaderTags;
    {
        std::map<std::string, std::string> readGroups;
        for (const Lane &L : lanes)
        {
            if (L.tiles.empty()) continue;                                      // a lane without data has no tiles
            std::string unit = o.bamPuFormat;
            const auto replace = [&unit](const std::string &what, const std::string &with) { for (size_t at = unit.find(what); std::string::npos != at; at = unit.find(what, at + with.size())) unit.replace(at, what.size(), with); };
            replace("%F", L.flowcell->flowcellId); replace("%L", std::to_string(L.lane->lane)); replace("%B", "none");
            readGroups[L.readGroup] = "@RG\tID:" + L.readGroup + "\tPL:ILLUMINA\tSM:default\tPU:" + unit;
        }
        for (const auto &rg : readGroups) headerLines.push_back(rg.second);
    }
    std::vector<const char *> linePointers, names, as, ur, m5;
    std::vector<uint32_t> lengths;
    for (const std::string &l : headerLines) linePointers.push_back(l.c_str());
    for (const isaac_reference_contig &c : reference.contigs)
    {
        names.push_back(c.name); lengths.push_back(uint32_t(c.total_bases)); as.push_back(c.bam_sq_as); ur.push_back(c.bam_sq_ur[0] ? c.bam_sq_ur : c.file); m5.push_back(c.bam_m5);
    }
    std::string commandLine;
    for (const std::string &a : o.argv) commandLine += (commandLine.empty() ? "" : " ") + a;
    uint64_t headerBytes = 0;
    isaac_gpu_bam_header(commandLine.c_str(), o.description.c_str(), VERSION, linePointers.data(), uint32_t(linePointers.size()), names.data(), lengths.data(), as.data(), ur.data(), m5.data(),
                         nContigs, 0, 0, &headerBytes);
    std::vector<uint8_t> header(headerBytes);
    if (isaac_gpu_bam_header(commandLine.c_str(), o.description.c_str(), VERSION, linePointers.data(), uint32_t(linePointers.size()), names.data(), lengths.data(), as.data(), ur.data(), m5.data(),
                             nContigs, header.data(), header.size(), &headerBytes)) throw std::runtime_error(std::string("isaac_gpu_bam_header: ") + isaac_gpu_bam_last_error());
    const auto compressOnHost = [&o](const uint8_t *data, uint64_t n, int eofBlock)
    {
        std::vector<uint8_t> out(isaac_gpu_bgzf_bound(n) + 64);
        uint64_t nOut = 0;
        if (isaac_gpu_bgzf_compress(data, n, o.bamGzipLevel, o.jobs, eofBlock, out.data(), out.size(), &nOut)) throw std::runtime_error(std::string("isaac_gpu_bgzf_compress: ") + isaac_gpu_bam_last_error());
        out.resize(nOut);
        return out;
    };
    const std::vector<uint8_t> headerBgzf = compressOnHost(header.data(), header.size(), 0);
    const std::vector<uint8_t> eofBlock = compressOnHost(0, 0, 1);

    // the bins in file order: the contigs, the unaligned templates behind them or (--keep-unaligned front) ahead of them
    std::vector<uint32_t> fileOrder;
    if ("front" == o.keepUnaligned) fileOrder.push_back(nBins - 1);
    for (uint32_t b = 0; b + 1 < nBins; ++b) fileOrder.push_back(b);
    if ("front" != o.keepUnaligned) fileOrder.push_back(nBins - 1);
    std::vector<BinOutput> outputs(fileOrder.size());
    if (pinnedWarm.joinable()) pinnedWarm.join();
    std::mutex outputLock; std::condition_variable outputReady, outputTaken;
    std::atomic<size_t> nextBin(0);
    size_t binsWrittenSoFar = 0;                        // (under outputLock) the builders stay at most this far ahead of the file: finished bins wait in host memory
    isaac_bam_options bamOptions; std::memset(&bamOptions, 0, sizeof(bamOptions));
    bamOptions.forced_dodgy_alignment_score = o.forcedDodgyAlignmentScore(); bamOptions.pessimistic_mapq = o.pessimisticMapQ; bamOptions.read_group = "0"; bamOptions.barcode = "none";
    bamOptions.mark_duplicates = o.markDuplicates; bamOptions.keep_duplicates = o.keepDuplicates; bamOptions.realign_gaps = "no" != o.realignGaps; bamOptions.realign_vigorously = o.realignVigorously; bamOptions.realign_dodgy = o.realignDodgy;
    bamOptions.bin_filter = 2;
    const uint32_t maxReadLength = std::max(params.read_length[0], params.read_length[1]);
    const bool syncDownloads = 0 != std::getenv("ISAAC_ALIGN_SYNC_DOWNLOADS");       // (measurements: a bin's blocks fetched before the next bin is begun)
    auto buildBins = [&](Worker &w)
    {
        const double start = seconds();
        DeviceMemory data, bam, bgzfSets[2], entries;
        unsigned turn = 0;                               // the blocks of one bin leave for the host (isaac_gpu_download_async) while the next bin is encoded into the other buffer
        std::vector<uint8_t> fromFile;                   // a spilled part on its way back
        struct Pending { bool active = false; size_t k = 0; uint64_t ticket = 0; BinOutput result; } pending;
        const auto publish = [&](size_t k, BinOutput &result)
        {
            result.ready = true;
            { std::lock_guard<std::mutex> guard(outputLock); outputs[k] = std::move(result); }
            outputReady.notify_all();
        };
        const auto finishPending = [&]()
        {
            if (!pending.active) return;
            const double waitStart = seconds();
            if (isaac_gpu_download_wait(w.ctx, pending.ticket) && pending.result.error.empty()) pending.result.error = std::string("isaac_gpu_download_wait: ") + isaac_gpu_last_error();
            w.downloadSeconds += seconds() - waitStart;
            publish(pending.k, pending.result);
            pending.active = false; pending.result = BinOutput();
        };
        for (size_t k = nextBin++; k < fileOrder.size(); k = nextBin++)
        {
            {   // not too far ahead of the writer (which may be waiting for the bin this worker still holds)
                const double waitStart = seconds();
                std::unique_lock<std::mutex> guard(outputLock);
                if (!(k < binsWrittenSoFar + BUILD_AHEAD)) { guard.unlock(); finishPending(); guard.lock(); }
                outputTaken.wait(guard, [&] { return k < binsWrittenSoFar + BUILD_AHEAD; });
                w.writerWaitSeconds += seconds() - waitStart;
            }
            DeviceMemory &bgzf = bgzfSets[turn];
            bool inFlight = false; uint64_t ticket = 0;
            BinOutput result;
            double mark = seconds();
            const auto lap = [&mark](double &into) { const double now = seconds(); into += now - mark; mark = now; };
            try
            {
                Bin &bin = bins[fileOrder[k]];
                if (!bin.parts.empty())
                {
                    // the bin's parts on this device, each the three arrays of a tile: parts in this worker's own blocks are used where they are;
                    // the others -- in host memory, or in the blocks of a worker on another device -- come into one buffer
                    uint64_t foreignBytes = 0;
                    for (const BinPart &part : bin.parts) if (!part.block || part.place != w.place) foreignBytes += part.bytes;
                    if (data.bytes() < foreignBytes) data.reset(w.ctx, foreignBytes);
                    std::vector<isaac_bam_tile> bamTiles(bin.parts.size());
                    uint64_t at = 0;
                    for (size_t i = 0; i < bin.parts.size(); ++i)
                    {
                        BinPart &part = bin.parts[i];
                        uint8_t *base = 0;
                        if (part.block && part.place == w.place) base = part.block->as<uint8_t>() + part.offset;
                        else
                        {
                            base = data.as<uint8_t>() + at;
                            // (from the other device on this context's own stream: nothing is asked of the context that owns the block, which is busy
                            // with bins of its own)
                            if (part.block) GPU(isaac_gpu_copy(w.ctx, base, part.block->as<uint8_t>() + part.offset, part.bytes));
                            else if (part.spilled)
                            {
                                if (fromFile.size() < part.bytes) fromFile.resize(part.bytes);
                                for (uint64_t done = 0; done < part.bytes; )
                                {
                                    const ssize_t r = ::pread(bin.spillFd, fromFile.data() + done, size_t(part.bytes - done), off_t(part.fileOffset + done));
                                    if (r <= 0) { if (r < 0 && EINTR == errno) continue; throw std::runtime_error("Failed to read " + bin.spillPath); }
                                    done += uint64_t(r);
                                }
                                GPU(isaac_gpu_upload(w.ctx, base, fromFile.data(), part.bytes));
                            }
                            else { GPU(isaac_gpu_upload(w.ctx, base, part.data.get(), part.bytes)); part.data.reset(); hostPartBytes -= part.bytes; }
                            at += part.bytes;
                        }
                        isaac_bam_tile &b = bamTiles[i];
                        b.bcl_dev = base;
                        b.fragments_dev = reinterpret_cast<const isaac_fragment *>(base + align64(part.clusters * clusterLength));
                        b.cigar_dev = reinterpret_cast<const uint32_t *>(base + align64(align64(part.clusters * clusterLength) + part.clusters * nReads * sizeof(isaac_fragment)));
                        b.n_records = part.clusters * nReads;
                        b.read_name_prefix = part.tile->namePrefix.c_str(); b.read_group = part.tile->readGroup.c_str(); b.tls = &part.tile->tls;
                    }
                    GPU(isaac_gpu_synchronize(w.ctx));
                    for (BinPart &part : bin.parts) if (part.block && part.place != w.place) part.block.reset();
                    lap(w.uploadSeconds);
                    isaac_bam_options options = bamOptions;
                    options.bin_first_position = bin.firstPosition; options.bin_end_position = bin.endPosition; options.bin_unaligned = bin.unaligned ? 1 : 0;
                    uint64_t capacity = bin.records * (96 + 2 * uint64_t(maxReadLength)), nBytes = 0, unalignedOffset = 0;
                    if (bam.bytes() < capacity) bam.reset(w.ctx, capacity);
                    if (entries.bytes() < bin.records * sizeof(isaac_bam_index_entry)) entries.reset(w.ctx, bin.records * sizeof(isaac_bam_index_entry));
                    options.index_entries_dev = entries.as<isaac_bam_index_entry>();
                    int rc = isaac_gpu_bam_records(w.ctx, bamTiles.data(), uint32_t(bamTiles.size()), &options, bam.as<uint8_t>(), bam.bytes(), &nBytes, &result.nRecords, &unalignedOffset);
                    if (ISAAC_GPU_ECAPACITY == rc)
                    {
                        bam.reset(w.ctx, nBytes);
                        rc = isaac_gpu_bam_records(w.ctx, bamTiles.data(), uint32_t(bamTiles.size()), &options, bam.as<uint8_t>(), bam.bytes(), &nBytes, &result.nRecords, &unalignedOffset);
                    }
                    check(rc, "isaac_gpu_bam_records");
                    lap(w.recordsSeconds);
                    if (nBytes)
                    {
                        // what the index wants to know about the records, which also says where one contig's records end and the next one's begin
                        result.entries = pinned.take(result.nRecords * sizeof(isaac_bam_index_entry));
                        GPU(isaac_gpu_download(w.ctx, result.entries.p, entries.as<isaac_bam_index_entry>(), result.nRecords * sizeof(isaac_bam_index_entry)));
                        lap(w.downloadSeconds);
                        const isaac_bam_index_entry *e = static_cast<const isaac_bam_index_entry *>(result.entries.p);
                        for (uint64_t i = 0; i < result.nRecords; )
                        {
                            uint64_t j = i + 1;
                            while (j < result.nRecords && e[j].ref_id == e[i].ref_id) ++j;
                            const uint64_t begin = e[i].offset, end = j < result.nRecords ? e[j].offset : nBytes;
                            result.segments.push_back(BinOutput::Segment{ 0, 0, begin, end - begin, i, j - i });
                            i = j;
                        }
                        // BGZF on the device, a run of blocks per contig: stored blocks at level 0 (bgzf::BgzfCompressor's own), deflated ones otherwise
                        uint64_t bound = 0;
                        for (const BinOutput::Segment &sg : result.segments) bound += align64(o.bamGzipLevel ? isaac_gpu_bgzf_deflate_bound(sg.recordsBytes) : isaac_gpu_bgzf_store_bound(sg.recordsBytes));
                        if (bgzf.bytes() < bound) bgzf.reset(w.ctx, bound);
                        uint64_t at = 0;
                        for (BinOutput::Segment &sg : result.segments)
                        {
                            uint64_t nOut = 0;
                            if (o.bamGzipLevel) GPU(isaac_gpu_bgzf_deflate(w.ctx, bam.as<uint8_t>() + sg.recordsOffset, sg.recordsBytes, 0, bgzf.as<uint8_t>() + at, bgzf.bytes() - at, &nOut));
                            else GPU(isaac_gpu_bgzf_store(w.ctx, bam.as<uint8_t>() + sg.recordsOffset, sg.recordsBytes, 0, bgzf.as<uint8_t>() + at, bgzf.bytes() - at, &nOut));
                            sg.bgzfOffset = at; sg.bgzfBytes = nOut;
                            at += nOut;                                    // (the runs lie back to back: the file is their concatenation)
                        }
                        lap(w.deflateSeconds);
                        result.bgzf = pinned.take(at); result.bgzfBytes = at; result.recordsBytes = nBytes;
                        finishPending();                                   // (the bin before this one: its blocks left while this one was encoded)
                        if (syncDownloads) GPU(isaac_gpu_download(w.ctx, result.bgzf.p, bgzf.as<uint8_t>(), at));
                        else { GPU(isaac_gpu_download_async(w.ctx, result.bgzf.p, bgzf.as<uint8_t>(), at, &ticket)); inFlight = true; }
                        lap(w.downloadSeconds);
                    }
                    w.noteMemory();
                    std::vector<BinPart>().swap(bin.parts);
                    bin.closeSpill();
                    lap(w.releaseSeconds);
                }
            }
            catch (const std::exception &e) { result.error = e.what(); }
            finishPending();                                               // bins are published in the order this worker took them
            if (inFlight) { pending.active = true; pending.k = k; pending.ticket = ticket; pending.result = std::move(result); turn ^= 1; }
            else publish(k, result);
        }
        finishPending();
        w.buildSeconds = seconds() - start;
    };
    std::vector<std::thread> builders;
    for (auto &w : workers) builders.emplace_back(buildBins, std::ref(*w));

    uint64_t nRecordsWritten = 0, binsWritten = 0;
    double writeSeconds = 0, indexBusySeconds = 0;
    std::string failure;
    {
        const int fd = outputFile.fd;
        uint64_t fileAt = 0;
        const auto append = [&](const uint8_t *data, uint64_t bytes) { if (fd >= 0 && bytes) { writeAt(fd, data, bytes, fileAt); fileAt += bytes; } };
        try { append(headerBgzf.data(), headerBgzf.size()); } catch (const std::exception &e) { failure = e.what(); }
        isaac_bam_indexer *indexer = isaac_gpu_bam_indexer_create(nContigs, headerBgzf.size());
        // the index takes its bins on a thread of its own, in the writer's order: going over 32 bytes and a BGZF block table per record costs about as much as
        // writing the record does (100 M pairs: 2 s beside 3 s of writes), and on the writer's thread it held the builders up
        std::mutex indexLock; std::condition_variable indexWake;
        std::deque<BinOutput> indexQueue; bool indexClosed = false; std::string indexFailure; double indexSeconds = 0;
        std::thread indexThread([&]()
        {
            for (;;)
            {
                BinOutput out;
                {
                    std::unique_lock<std::mutex> guard(indexLock);
                    indexWake.wait(guard, [&] { return indexClosed || !indexQueue.empty(); });
                    if (indexQueue.empty()) return;
                    out = std::move(indexQueue.front()); indexQueue.pop_front();
                }
                indexWake.notify_all();
                const double start = seconds();
                isaac_bam_index_entry *entries = static_cast<isaac_bam_index_entry *>(out.entries.p);
                for (const BinOutput::Segment &sg : out.segments)
                {
                    if (!indexFailure.empty()) break;
                    for (uint64_t i = sg.firstEntry; i < sg.firstEntry + sg.nEntries; ++i) entries[i].offset -= sg.recordsOffset;      // offsets inside the run's own records
                    if (isaac_gpu_bam_indexer_add_entries(indexer, entries + sg.firstEntry, sg.nEntries, sg.recordsBytes, static_cast<const uint8_t *>(out.bgzf.p) + sg.bgzfOffset, sg.bgzfBytes))
                        indexFailure = std::string("isaac_gpu_bam_indexer_add_entries: ") + isaac_gpu_bam_index_last_error();
                }
                indexSeconds += seconds() - start;
                pinned.give(out.bgzf); pinned.give(out.entries);
            }
        });
        struct CloseIndex { std::mutex &lock; std::condition_variable &wake; bool &closed; std::thread &t; void operator()() { { std::lock_guard<std::mutex> g(lock); closed = true; } wake.notify_all(); if (t.joinable()) t.join(); } ~CloseIndex() { (*this)(); } }
            closeIndex{ indexLock, indexWake, indexClosed, indexThread };
        for (size_t k = 0; k < outputs.size(); ++k)
        {
            BinOutput out;
            {
                std::unique_lock<std::mutex> guard(outputLock);
                outputReady.wait(guard, [&] { return outputs[k].ready; });
                out = std::move(outputs[k]);
                binsWrittenSoFar = k + 1;
            }
            outputTaken.notify_all();
            if (!out.error.empty() && failure.empty()) failure = out.error;
            if (failure.empty() && out.bgzfBytes)
            {
                const double writeStart = seconds();
                try { append(static_cast<const uint8_t *>(out.bgzf.p), out.bgzfBytes); } catch (const std::exception &e) { failure = e.what(); }
                writeSeconds += seconds() - writeStart;
                nRecordsWritten += out.nRecords; ++binsWritten;
                if (failure.empty())
                {   // to the index, which keeps the buffers until it is through with them (at most four bins behind the file)
                    std::unique_lock<std::mutex> guard(indexLock);
                    indexWake.wait(guard, [&] { return indexQueue.size() < 4; });
                    indexQueue.push_back(std::move(out));
                    guard.unlock();
                    indexWake.notify_all();
                    continue;
                }
            }
            pinned.give(out.bgzf); pinned.give(out.entries);
        }
        for (std::thread &t : builders) t.join();
        closeIndex();
        if (failure.empty()) failure = indexFailure;
        indexBusySeconds = indexSeconds;
        if (failure.empty()) try { append(eofBlock.data(), eofBlock.size()); } catch (const std::exception &e) { failure = e.what(); }
        preallocator.finish();
        if (::ftruncate(fd, off_t(fileAt)) && failure.empty()) failure = "Failed to write " + bamPath;       // (gives back what was asked for beyond the end)
        outputFile.fd = -1;
        if (::close(fd) && failure.empty()) failure = "Failed to write " + bamPath;
        if (failure.empty())
        {
            uint64_t baiBytes = 0;
            isaac_gpu_bam_indexer_finish(indexer, 0, 0, &baiBytes);
            std::vector<uint8_t> bai(baiBytes);
            if (isaac_gpu_bam_indexer_finish(indexer, bai.data(), bai.size(), &baiBytes)) failure = std::string("isaac_gpu_bam_indexer_finish: ") + isaac_gpu_bam_index_last_error();
            else
            {
                std::ofstream index((bamPath + ".bai").c_str(), std::ios::binary | std::ios::trunc);
                if (!index || !index.write(reinterpret_cast<const char *>(bai.data()), std::streamsize(bai.size()))) failure = "Error opening bam index file for writing " + bamPath + ".bai";
            }
        }
        isaac_gpu_bam_indexer_destroy(indexer);
    }
    if (!failure.empty()) throw std::runtime_error(failure);
    const double buildSeconds = seconds() - buildStart, total = seconds() - runStart;
    // the bins' ranges (ReferencePosition values) for whoever checks the file against the same plan
    std::string binRangesJson = "[";
    for (size_t b = 0; b < plan.ranges.size(); ++b) binRangesJson += (b ? ", [" : "[") + std::to_string(plan.ranges[b].first) + ", " + std::to_string(plan.ranges[b].second) + "]";
    binRangesJson += "]";
    // device time of the build stage's launch sequences on the first worker (HIP events on its stream: isaac_gpu_kernel_time_ms), in seconds over the run
    std::string buildKernelsJson = "{";
    for (const char *name : { "bam_duplicates", "bam_realign", "bam_order", "bam_encode", "bgzf_deflate", "bgzf_store" })
    {
        double ms = 0; uint64_t launches = 0;
        if (isaac_gpu_kernel_time_ms(workers[0]->ctx, name, &ms, &launches) || !launches) continue;
        buildKernelsJson += std::string(buildKernelsJson.size() > 1 ? ", \"" : "\"") + name + "\": " + std::to_string(ms * double(launches) / 1000.0);
    }
    buildKernelsJson += "}";
    uint64_t tilesOnDevice = 0, loadsOnDevice = 0, peakDevice = 0, nLoads = 0; double selectBusySeconds = 0;
    for (auto &w : workers) selectBusySeconds += w->selectSeconds;
    for (auto &w : workers) { tilesOnDevice += w->tilesKeptOnDevice; loadsOnDevice += w->loadsKeptOnDevice; peakDevice = std::max(peakDevice, w->peakDeviceBytes); }
    for (const Lane &L : lanes) nLoads += L.loads.size();
    std::cerr << "isaac-align: " << bamPath << ": " << nRecordsWritten << " records in " << binsWritten << " bin(s)" << std::endl;
    // one line for scripts (bench.py): what the run took, stage by stage
    std::cerr << "isaac-align: timing {\"clusters\": " << totalClusters << ", \"reads\": " << totalClusters * nReads << ", \"records\": " << nRecordsWritten << ", \"workers\": " << workers.size()
              << ", \"reference_s\": " << referenceSeconds << ", \"reference_fasta_s\": " << fastaSeconds << ", \"reference_contigs_s\": " << contigSeconds << ", \"reference_table_s\": " << tableSeconds << ", \"reference_share_s\": " << shareSeconds << ", \"load_and_find_s\": " << loadSeconds << ", \"load_text_wait_s\": " << g_textWaitSeconds << ", \"load_convert_s\": " << g_convertSeconds << ", \"load_first_lookup_s\": " << g_firstLookupSeconds << ", \"load_text_open_s\": " << g_textOpenSeconds << ", \"load_memory_s\": " << g_loadMemorySeconds << ", \"load_place_s\": " << g_loadPlaceSeconds << ", \"select_resolve_s\": " << g_resolveSeconds << ", \"select_and_bin_s\": " << selectSeconds << ", \"select_busy_s\": " << selectBusySeconds << ", \"selection_streamed\": " << (streamSelection ? 1 : 0) << ", \"build_and_write_s\": " << buildSeconds
              << ", \"tiles_kept_on_device\": " << tilesOnDevice << ", \"tiles\": " << tiles.size() << ", \"loads_kept_on_device\": " << loadsOnDevice << ", \"loads\": " << nLoads << ", \"bins\": " << nBins << ", \"bin_cuts\": " << plan.cuts.size() << ", \"estimated_clusters\": " << estimatedClusters << ", \"bin_ranges\": " << binRangesJson
              << ", \"build_upload_s\": " << workers[0]->uploadSeconds << ", \"build_records_s\": " << workers[0]->recordsSeconds << ", \"build_deflate_s\": " << workers[0]->deflateSeconds
              << ", \"build_download_s\": " << workers[0]->downloadSeconds  << ", \"build_writer_wait_s\": " << workers[0]->writerWaitSeconds << ", \"build_release_s\": " << workers[0]->releaseSeconds << ", \"build_device_s\": " << buildKernelsJson << ", \"spilled_bytes\": " << spilledBytes.load() << ", \"preallocated_bytes\": " << preallocator.done() << ", \"file_write_s\": " << writeSeconds << ", \"index_s\": " << indexBusySeconds
              << ", \"overflow_clusters\": " << overflowClusters << ", \"mapq_near_integer\": " << mapqNearInteger << ", \"mapq_resolved_on_host\": " << mapqResolved << ", \"mapq_changed_by_host\": " << mapqChanged
              << ", \"peak_device_bytes\": " << peakDevice << ", \"peak_host_bytes\": " << hostResidentBytes()
              << ", \"total_s\": " << total << "}" << std::endl;
    int status = 0;
    if (overflowClusters)
    {   // a fixed work list of the device was too small for these clusters: their records are flagged (isaac_fragment::reserved bit 2) and not exact
        std::cerr << "ERROR: " << overflowClusters << " cluster(s) exceeded a fixed work list of the device; the records of these clusters in " << bamPath << " are not what the reference writes" << std::endl;
        status = 3;
    }
    // Everything the run was asked for is in its files, and every thread it started has ended.  What is left is giving memory back -- some 180 GB of device
    // memory one hipFree at a time, 60 GB of page-locked and plain host memory -- which the operating system does for a process that ends in a fraction of the
    // time the destructors take (1.5 s of a 15 s run).  ISAAC_ALIGN_ORDERLY_EXIT=1: the destructors run (leak checkers).
    if (!std::getenv("ISAAC_ALIGN_ORDERLY_EXIT")) { std::cerr.flush(); std::clog.flush(); std::cout.flush(); std::_Exit(status); }
    return status;
}

} // namespace

int main(int argc, char **argv)
{
    try
    {
        const AlignOptions options = AlignOptions::parse(argc, argv);
        if (AlignOptions::HELP == options.action) { std::cout << AlignOptions::usage() << std::endl; return 0; }
        if (AlignOptions::VERSION == options.action) { std::cout << VERSION << std::endl; return 0; }
        return run(options);
    }
    catch (const InvalidOption &e)
    {   // common::run: the message, then the hint, exit code 1 (include/common/Program.hh:60-92)
        std::clog << "Failed to parse the options: " << e.what() << std::endl << "Use --help for the options this host takes." << std::endl;
        return 1;
    }
    catch (const std::exception &e)
    {
        std::clog << "isaac-align: " << e.what() << std::endl;
        return 1;
    }
}
