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
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <sys/stat.h>
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

Reference loadReference(const std::string &xmlPath)
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
struct Tile
{
    unsigned lane = 0, number = 0, index = 0, clusters = 0;     // tile number within the lane (the read name), index over the run (the records)
    const uint8_t *bcl = 0;                                       // inside its load's buffer, on its worker's device
    Worker *worker = 0;
    isaac_tls tls;
    std::string namePrefix, readGroup;
};

// one read of one lane: the file and the text not yet converted
const size_t TEXT_CHUNK = size_t(64) << 20;
// where the load phase's time goes (one thread runs it): waiting for text, upload + conversion, first lookup
double g_textWaitSeconds = 0, g_convertSeconds = 0, g_firstLookupSeconds = 0;

// (A thread per read that fetched the next piece ahead of the conversion was measured and taken out again: 1.76 s against 1.3 s for 6.6 GB of text -- the
// reader's fresh buffers and the extra copy cost more than the overlap brought; profiles/r4_e_cli_timing.log.)
struct ReadStream
{
    std::unique_ptr<FastqFileReader> reader;
    // the text not yet converted, [begin, end) of a page-locked buffer: the file is read into it and the device takes it from there at the link's
    // rate (a std::vector is cleared before it is read into and uploaded through the runtime's own staging: half of the load phase's time)
    char *text = 0; size_t capacity = 0, begin = 0, end = 0;
    uint64_t consumedBytes = 0;                                   // of the uncompressed text, for error messages
    bool ended = false;                                           // the reader was at the end of its file when last looked at
    ReadStream() {}
    ReadStream(const ReadStream &) = delete;
    ReadStream &operator=(const ReadStream &) = delete;
    ~ReadStream() { if (text) isaac_gpu_host_free(text); }
    size_t size() const { return end - begin; }
    const char *data() const { return text + begin; }
    void consume(size_t n) { begin += n; if (begin == end) begin = end = 0; }
    // more text behind what is there, up to `want` bytes in all if the file has them
    void fill(size_t want)
    {
        if (size() >= want || ended) return;
        if (capacity < want)
        {
            void *larger = 0;
            GPU(isaac_gpu_host_malloc(want, &larger));
            if (size()) std::memcpy(larger, data(), size());
            if (text) isaac_gpu_host_free(text);
            text = static_cast<char *>(larger); capacity = want; end = size(); begin = 0;
        }
        else if (begin) { std::memmove(text, text + begin, size()); end = size(); begin = 0; }
        end += reader->readInto(text + end, want - end);
        ended = reader->atEnd();
    }
    bool atEnd() const { return ended; }
};

// io::FastqLoader::loadSingleRead for up to maxClusters clusters: the text goes to the device in pieces, the converter leaves the incomplete
// record at the end of a piece for the next one
uint32_t loadRead(isaac_gpu_ctx *ctx, ReadStream &stream, unsigned readIndex, bool allowVariableLength, uint8_t *bclDev, unsigned clusterLength, uint32_t maxClusters, DeviceMemory &textDev)
{
    uint32_t clusters = 0;
    size_t chunk = TEXT_CHUNK;
    while (clusters < maxClusters)
    {
        const double fillStart = seconds();
        stream.fill(chunk);
        g_textWaitSeconds += seconds() - fillStart;
        if (!stream.size()) break;
        const double convertStart = seconds();
        const bool final = stream.atEnd();
        if (textDev.bytes() < stream.size() + 64) textDev.reset(ctx, stream.size() + 64);
        GPU(isaac_gpu_upload(ctx, textDev.as<char>(), stream.data(), stream.size()));
        uint32_t n = 0; uint64_t consumed = 0, errorOffset = 0;
        const int rc = isaac_gpu_fastq_to_bcl(ctx, textDev.as<char>(), stream.size(), readIndex, allowVariableLength, final, bclDev + uint64_t(clusters) * clusterLength,
                                              maxClusters - clusters, &n, &consumed, &errorOffset);
        if (rc)
            throw std::runtime_error(stream.reader->path() + ": " + isaac_gpu_last_error() + " (record " + std::to_string(clusters + n) + " of this load, offset " +
                                     std::to_string(stream.consumedBytes + errorOffset) + ")");
        g_convertSeconds += seconds() - convertStart;
        clusters += n;
        stream.consumedBytes += consumed;
        stream.consume(consumed);
        if (!n && !consumed)
        {
            if (final) break;                                     // nothing but line ends left
            chunk *= 2;                                           // a record longer than the piece
        }
    }
    return clusters;
}

// ---- the bins: what BinningFragmentStorage keeps in files (lib/alignment/matchSelector/BinningFragmentStorage.cpp) kept in host memory ----------
// One bin per contig and one for the templates without a position: the bins of the BAM stage (duplicates and realignment never look across
// contigs).  A tile leaves one part in every bin it has records in (isaac_gpu_bin_tile): BCL bytes, records and CIGAR words of the clusters
// concerned, about 150 + 2 x read length bytes per read.  That is the run's memory model: HBM holds the table, the BCL bytes of the loads
// until their tiles are selected, one tile's scratch and -- in the build stage -- one bin at a time.  The bins' parts stay where isaac_gpu_bin_tile
// wrote them, a block of device memory per tile, for as long as the device has room beside what the build stage will want (a quarter of its
// memory is left alone); the tiles after that leave their parts in host memory.  A part on the device goes into its bin's BAM stage as it
// lies there, no copy in either direction: on a 288 GB device that is every run of up to some 250 million pairs.
struct BinPart { const Tile *tile; uint64_t clusters, words, bytes; std::unique_ptr<uint8_t[]> data; std::shared_ptr<DeviceMemory> block; uint64_t offset = 0; int device = -1; };
struct Bin { std::mutex lock; std::vector<BinPart> parts; uint64_t bytes = 0, records = 0; };

uint64_t align64(uint64_t v) { return (v + 63) & ~uint64_t(63); }

// a device and what runs on it
struct Worker
{
    int device = 0; unsigned id = 0;
    isaac_gpu_ctx *ctx = 0;
    std::vector<DeviceMemory> loads;                // the BCL bytes of the loads dealt to this worker
    std::vector<Tile *> tiles;
    DeviceMemory matches, offsets, textDev;
    uint64_t matchCapacity = 0;
    isaac_counters counters;
    uint64_t tilesKeptOnDevice = 0;
    double selectSeconds = 0, buildSeconds = 0, uploadSeconds = 0, recordsSeconds = 0, deflateSeconds = 0, downloadSeconds = 0;
    ~Worker() { loads.clear(); matches.release(); offsets.release(); textDev.release(); if (ctx) isaac_gpu_destroy(ctx); }
};

// what a bin of the file becomes: its BGZF blocks, and what the index wants to know about its records (buffers that are not cleared first:
// they are gigabytes)
struct BinOutput
{
    bool ready = false; std::unique_ptr<uint8_t[]> bgzf; std::unique_ptr<isaac_bam_index_entry[]> entries; uint64_t bgzfBytes = 0, recordsBytes = 0, nRecords = 0; std::string error;
};

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

    // ---- the workers and the reference: the first context of a device loads the table, the others of that device share it; another device
    // gets a copy over the link between the two (isaac_gpu_copy)
    const std::vector<int> devices = o.deviceList();
    std::vector<std::unique_ptr<Worker> > workers;
    Reference reference;
    double fastaSeconds = 0, contigSeconds = 0, tableSeconds = 0;
    {
        Stage stage("loading the reference");
        reference = loadReference(o.referenceGenome);
        fastaSeconds = seconds() - runStart;
        for (size_t k = 0; k < devices.size(); ++k)
        {
            workers.emplace_back(new Worker);
            Worker &w = *workers.back();
            w.device = devices[k]; w.id = unsigned(k);
            GPU(isaac_gpu_create(w.device, &params, ISAAC_GPU_STREAM_OWN, &w.ctx));
            const double contigStart = seconds();
            GPU(isaac_gpu_load_contigs(w.ctx, reference.bases.get(), reference.offsets.data(), uint32_t(reference.contigs.size())));
            contigSeconds += seconds() - contigStart;
            if (0 == k) { const double tableStart = seconds(); GPU(isaac_gpu_load_sorted_reference(w.ctx, o.referenceGenome.c_str())); tableSeconds = seconds() - tableStart; continue; }
            // the first worker of a device that is not the first worker's gets a copy of the table, everybody else reads one that is there
            Worker *sameDevice = 0;
            for (size_t j = 0; j < k && !sameDevice; ++j) if (workers[j]->device == w.device) sameDevice = workers[j].get();
            GPU(isaac_gpu_share_index(w.ctx, sameDevice ? sameDevice->ctx : workers[0]->ctx));
        }
        reference.bases.reset();
    }
    const uint32_t nContigs = uint32_t(reference.contigs.size());
    if (nContigs + 1 > 255) throw std::runtime_error("this host keeps one bin per contig: at most 254 contigs");
    const double referenceSeconds = seconds() - runStart;

    // ---- FastqSeedSource: loads of --clusters-at-a-time clusters, tiles of at most tileClustersMax; the loads are dealt to the workers in turn
    const uint32_t tileClustersMax = isaac_gpu_fastq_tile_clusters_max(o.clustersAtATime, params.n_seeds);
    const uint32_t loadClusters = o.clustersAtATime ? o.clustersAtATime : 4 * tileClustersMax;      // a multiple of the tile size: the tiles come out the same for any such load
    std::deque<Tile> tiles;
    std::vector<uint8_t> contigHasMatches(nContigs, 0);
    auto findMatches = [&](Worker &w, const Tile &t, uint64_t &nMatches, uint8_t *hits)
    {
        const uint64_t worst = uint64_t(t.clusters) * 2 * params.n_seeds * std::max(1u, params.repeat_threshold - 1);
        if (!w.offsets.bytes()) w.offsets.reset(w.ctx, (uint64_t(tileClustersMax) + 1) * 8);
        if (!w.matchCapacity) { w.matchCapacity = std::max<uint64_t>(1024, std::min<uint64_t>(worst, uint64_t(tileClustersMax) * 24)); w.matches.reset(w.ctx, w.matchCapacity * sizeof(isaac_match)); }
        for (;;)
        {
            const int rc = isaac_gpu_find_matches(w.ctx, t.bcl, t.clusters, t.index, w.matches.as<isaac_match>(), w.matchCapacity, w.offsets.as<uint64_t>(), &nMatches, hits);
            if (ISAAC_GPU_ECAPACITY != rc) { check(rc, "isaac_gpu_find_matches"); return; }
            w.matchCapacity = std::max(nMatches, 2 * w.matchCapacity);
            w.matches.reset(w.ctx, w.matchCapacity * sizeof(isaac_match));
        }
    };
    uint64_t totalClusters = 0;
    const double loadStart = seconds();
    {
        Stage stage("loading base calls and finding matches");
        unsigned barcodeIndex = 0;
        size_t nextWorker = 0;
        for (const FastqFlowcell &fc : flowcells)
            for (const FastqLane &lane : fc.lanes)
            {
                ReadStream streams[2];
                for (unsigned r = 0; r < nReads; ++r)
                {
                    if (lane.readPath[r].empty()) throw std::runtime_error("lane " + std::to_string(lane.lane) + " of " + fc.baseCallsDirectory + " has no read " + std::to_string(r + 1));
                    streams[r].reader.reset(new FastqFileReader(lane.readPath[r], fc.compressed));
                }
                const std::string readGroup = std::to_string(barcodeIndex++);         // one 'none' barcode per lane, numbered in the order of the lanes
                uint32_t nextTile = 1;
                for (;;)
                {
                    Worker &w = *workers[nextWorker % workers.size()];
                    DeviceMemory bcl(w.ctx, uint64_t(loadClusters) * clusterLength + 64);
                    uint32_t loaded[2] = { 0, 0 };
                    for (unsigned r = 0; r < nReads; ++r)
                        loaded[r] = loadRead(w.ctx, streams[r], r, o.variableReadLength || o.variableFastqReadLength, bcl.as<uint8_t>(), clusterLength, loadClusters, w.textDev);
                    if (2 == nReads && loaded[0] != loaded[1])
                        throw std::runtime_error("Mismatching number of clusters in " + lane.readPath[0] + " (" + std::to_string(loaded[0]) + ") and " + lane.readPath[1] + " (" + std::to_string(loaded[1]) + ")");
                    if (!loaded[0]) break;
                    if (loaded[0] < loadClusters / 2)
                    {   // "allocated too much memory for bcl data": the load keeps what it uses
                        DeviceMemory exact(w.ctx, uint64_t(loaded[0]) * clusterLength + 64);
                        GPU(isaac_gpu_copy(w.ctx, exact.as<uint8_t>(), bcl.as<uint8_t>(), uint64_t(loaded[0]) * clusterLength));
                        GPU(isaac_gpu_synchronize(w.ctx));
                        bcl = std::move(exact);
                    }
                    uint32_t nTiles = 0, next = 0;
                    isaac_gpu_fastq_tiles(loaded[0], o.clustersAtATime, params.n_seeds, nextTile, 0, 0, 0, &nTiles, &next);
                    std::vector<uint32_t> numbers(nTiles), sizes(nTiles);
                    GPU(isaac_gpu_fastq_tiles(loaded[0], o.clustersAtATime, params.n_seeds, nextTile, numbers.data(), sizes.data(), nTiles, &nTiles, &next));
                    nextTile = next;
                    uint64_t first = 0;
                    for (uint32_t k = 0; k < nTiles; ++k)
                    {
                        tiles.emplace_back();
                        Tile &t = tiles.back();
                        t.lane = lane.lane; t.number = numbers[k]; t.index = unsigned(tiles.size() - 1); t.clusters = sizes[k]; t.bcl = bcl.as<uint8_t>() + first * clusterLength; t.worker = &w;
                        t.namePrefix = fc.flowcellId + ":" + std::to_string(lane.lane) + ":" + std::to_string(t.number) + ":"; t.readGroup = readGroup;
                        std::memset(&t.tls, 0, sizeof(t.tls));
                        first += sizes[k];
                        uint64_t nMatches = 0;
                        const double lookupStart = seconds();
                        findMatches(w, t, nMatches, contigHasMatches.data());
                        g_firstLookupSeconds += seconds() - lookupStart;
                        w.tiles.push_back(&t);
                    }
                    totalClusters += loaded[0];
                    w.loads.push_back(std::move(bcl));
                    ++nextWorker;
                    if (loaded[0] < loadClusters) break;
                }
            }
        for (auto &w : workers) w->textDev.release();
        std::cerr << "isaac-align: " << totalClusters << " clusters in " << tiles.size() << " tile(s) on " << workers.size() << " worker(s)" << std::endl;
        if (tiles.empty()) throw InvalidOption("No data found to process. Please check your --base-calls.");
    }
    const double loadSeconds = seconds() - loadStart;

    // ---- SelectMatchesTransition: every tile with the contigs the whole run has matches on.  The template length statistics of a lane are
    // learnt tile by tile until a tile gives stable ones, which then serve the rest of the lane (MatchSelector.cpp:395-412): settled first, in tile
    // order, so that the workers can take their tiles in any order afterwards.
    std::vector<Bin> bins(nContigs + 1);
    std::vector<uint32_t> binOfContig(nContigs);
    for (uint32_t c = 0; c < nContigs; ++c) binOfContig[c] = c;
    const double selectStart = seconds();
    {
        Stage stage("selecting matches");
        for (auto &w : workers) GPU(isaac_gpu_set_loaded_contigs(w->ctx, contigHasMatches.data(), nContigs));
        {
            isaac_tls tls; std::memset(&tls, 0, sizeof(tls));
            std::string laneKey;
            for (Tile &t : tiles)
            {
                if (laneKey != t.readGroup) { std::memset(&tls, 0, sizeof(tls)); laneKey = t.readGroup; }       // barcodeTemplateLengthStatistics: one per barcode
                if (!tls.stable || o.perTileTls)
                {
                    uint64_t nMatches = 0;
                    findMatches(*t.worker, t, nMatches, 0);
                    GPU(isaac_gpu_determine_tls(t.worker->ctx, t.bcl, t.clusters, t.index, t.worker->matches.as<isaac_match>(), t.worker->offsets.as<uint64_t>(), &tls));
                    std::cerr << "isaac-align: template length statistics of tile " << t.namePrefix << " min " << tls.min << " median " << tls.median << " max " << tls.max
                              << (tls.stable ? " (stable)" : " (unstable)") << std::endl;
                }
                t.tls = tls;
            }
        }
        std::vector<std::string> errors(workers.size());
        const bool hostBins = 0 != std::getenv("ISAAC_ALIGN_HOST_BINS");          // tests: every part through host memory
        auto selectTiles = [&](Worker &w)
        {
            try
            {
                const double start = seconds();
                DeviceMemory slots(w.ctx, uint64_t(tileClustersMax) * nReads * ISAAC_GPU_MAX_CIGAR_OPS * 4), records(w.ctx, uint64_t(tileClustersMax) * nReads * sizeof(isaac_fragment)), packed, binned;
                std::vector<isaac_bin_size> sizes(nContigs + 1);
                for (Tile *tp : w.tiles)
                {
                    Tile &t = *tp;
                    uint64_t nMatches = 0;
                    findMatches(w, t, nMatches, 0);
                    const uint64_t nRecords = uint64_t(t.clusters) * nReads;
                    GPU(isaac_gpu_select_n(w.ctx, t.bcl, t.clusters, t.index, w.matches.as<isaac_match>(), nMatches, w.offsets.as<uint64_t>(), &t.tls, records.as<isaac_fragment>(), slots.as<uint32_t>(),
                                           nRecords * ISAAC_GPU_MAX_CIGAR_OPS));
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
                    // BinningFragmentStorage: the tile's clusters to their bins, every bin's part to host memory
                    uint64_t need = 0;
                    const uint64_t guess = align64(uint64_t(t.clusters) * clusterLength + nRecords * sizeof(isaac_fragment) + words * 4) * 5 / 4 + 256 * (nContigs + 1);
                    if (binned.bytes() < guess) binned.reset(w.ctx, guess);
                    rc = isaac_gpu_bin_tile(w.ctx, t.bcl, records.as<isaac_fragment>(), packed.as<uint32_t>(), t.clusters, binOfContig.data(), nContigs, nContigs + 1, binned.as<uint8_t>(), binned.bytes(),
                                            sizes.data(), &need);
                    if (ISAAC_GPU_ECAPACITY == rc)
                    {
                        binned.reset(w.ctx, need);
                        rc = isaac_gpu_bin_tile(w.ctx, t.bcl, records.as<isaac_fragment>(), packed.as<uint32_t>(), t.clusters, binOfContig.data(), nContigs, nContigs + 1, binned.as<uint8_t>(), binned.bytes(),
                                                sizes.data(), &need);
                    }
                    check(rc, "isaac_gpu_bin_tile");
                    // the tile's parts stay on the device while it has room (see BinPart)
                    std::shared_ptr<DeviceMemory> block;
                    {
                        uint64_t freeBytes = 0, totalBytes = 0;
                        GPU(isaac_gpu_memory_info(w.ctx, &freeBytes, &totalBytes));
                        if (!hostBins && freeBytes > totalBytes / 4 + binned.bytes()) { block = std::make_shared<DeviceMemory>(std::move(binned)); ++w.tilesKeptOnDevice; }
                    }
                    const uint8_t *binnedBytes = block ? block->as<uint8_t>() : binned.as<uint8_t>();
                    uint64_t at = 0;
                    for (uint32_t b = 0; b <= nContigs; ++b)
                    {
                        const uint64_t m = sizes[b].n_clusters, cw = sizes[b].n_cigar_words;
                        const uint64_t bytes = align64(align64(m * clusterLength) + m * nReads * sizeof(isaac_fragment)) + align64(cw * 4);
                        if (m)
                        {
                            BinPart part; part.tile = &t; part.clusters = m; part.words = cw; part.bytes = bytes;
                            if (block) { part.block = block; part.offset = at; part.device = w.device; }
                            else
                            {
                                part.data.reset(new uint8_t[bytes]);
                                GPU(isaac_gpu_download(w.ctx, part.data.get(), binnedBytes + at, bytes));
                            }
                            std::lock_guard<std::mutex> guard(bins[b].lock);
                            bins[b].bytes += bytes; bins[b].records += m * nReads;
                            bins[b].parts.push_back(std::move(part));
                        }
                        at += bytes;
                    }
                }
                GPU(isaac_gpu_synchronize(w.ctx));
                isaac_gpu_get_counters(w.ctx, &w.counters);
                w.loads.clear();                        // the BCL bytes are in the bins now
                w.matches.release(); w.offsets.release();
                w.selectSeconds = seconds() - start;
            }
            catch (const std::exception &e) { errors[w.id] = e.what(); }
        };
        std::vector<std::thread> threads;
        for (size_t k = 1; k < workers.size(); ++k) threads.emplace_back(selectTiles, std::ref(*workers[k]));
        selectTiles(*workers[0]);
        for (std::thread &t : threads) t.join();
        for (const std::string &e : errors) if (!e.empty()) throw std::runtime_error(e);
        uint64_t overflow = 0;
        for (auto &w : workers) overflow += w->counters.overflow_clusters;
        if (overflow) std::cerr << "WARNING: " << overflow << " cluster(s) exceeded a fixed work list; their records are flagged (isaac_fragment::reserved bit 2)" << std::endl;
        // parts in tile order inside a bin, whichever worker was first
        for (Bin &bin : bins) std::sort(bin.parts.begin(), bin.parts.end(), [](const BinPart &a, const BinPart &b) { return a.tile->index < b.tile->index; });
    }
    const double selectSeconds = seconds() - selectStart;

    // ---- build::Build: one bin at a time -- records, duplicates, realignment, BAM records, BGZF blocks on the device -- the file and its index
    // in bin order
    const double buildStart = seconds();
    Stage stage("building and writing sorted.bam");
    // header (Bam.hh:153-235): --bam-header-tag lines, the read groups in the order of a map keyed by their ids, the contigs in karyotype order
    std::vector<std::string> headerLines = o.bamHeaderTags;
    {
        std::map<std::string, std::string> readGroups;
        unsigned barcodeIndex = 0;
        for (const FastqFlowcell &fc : flowcells)
            for (const FastqLane &lane : fc.lanes)
            {
                const std::string id = std::to_string(barcodeIndex++);
                if (tiles.end() == std::find_if(tiles.begin(), tiles.end(), [&id](const Tile &t) { return t.readGroup == id; })) continue;       // a lane without data has no tiles
                std::string unit = o.bamPuFormat;
                const auto replace = [&unit](const std::string &what, const std::string &with) { for (size_t at = unit.find(what); std::string::npos != at; at = unit.find(what, at + with.size())) unit.replace(at, what.size(), with); };
                replace("%F", fc.flowcellId); replace("%L", std::to_string(lane.lane)); replace("%B", "none");
                readGroups[id] = "@RG\tID:" + id + "\tPL:ILLUMINA\tSM:default\tPU:" + unit;
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
    if ("front" == o.keepUnaligned) fileOrder.push_back(nContigs);
    for (uint32_t c = 0; c < nContigs; ++c) fileOrder.push_back(c);
    if ("front" != o.keepUnaligned) fileOrder.push_back(nContigs);
    std::vector<BinOutput> outputs(fileOrder.size());
    std::mutex outputLock; std::condition_variable outputReady;
    std::atomic<size_t> nextBin(0);
    isaac_bam_options bamOptions; std::memset(&bamOptions, 0, sizeof(bamOptions));
    bamOptions.forced_dodgy_alignment_score = o.forcedDodgyAlignmentScore(); bamOptions.pessimistic_mapq = o.pessimisticMapQ; bamOptions.read_group = "0"; bamOptions.barcode = "none";
    bamOptions.mark_duplicates = o.markDuplicates; bamOptions.keep_duplicates = o.keepDuplicates; bamOptions.realign_gaps = "no" != o.realignGaps; bamOptions.realign_dodgy = o.realignDodgy;
    bamOptions.bin_filter = 1;
    const uint32_t maxReadLength = std::max(params.read_length[0], params.read_length[1]);
    auto buildBins = [&](Worker &w)
    {
        const double start = seconds();
        DeviceMemory data, bam, bgzf, entries;
        for (size_t k = nextBin++; k < fileOrder.size(); k = nextBin++)
        {
            BinOutput result;
            double mark = seconds();
            const auto lap = [&mark](double &into) { const double now = seconds(); into += now - mark; mark = now; };
            try
            {
                Bin &bin = bins[fileOrder[k]];
                if (!bin.parts.empty())
                {
                    // the bin's parts to the device, each the three arrays of a tile
                    // parts that lie on this device are used where they are; the others -- in host memory, or on another worker's device --
                    // come into one buffer
                    uint64_t foreignBytes = 0;
                    for (const BinPart &part : bin.parts) if (!part.block || part.device != w.device) foreignBytes += part.bytes;
                    if (data.bytes() < foreignBytes) data.reset(w.ctx, foreignBytes);
                    std::vector<isaac_bam_tile> bamTiles(bin.parts.size());
                    std::unique_ptr<uint8_t[]> staging; uint64_t stagingBytes = 0;
                    uint64_t at = 0;
                    for (size_t i = 0; i < bin.parts.size(); ++i)
                    {
                        BinPart &part = bin.parts[i];
                        uint8_t *base = 0;
                        if (part.block && part.device == w.device) base = part.block->as<uint8_t>() + part.offset;
                        else
                        {
                            base = data.as<uint8_t>() + at;
                            if (part.block)
                            {   // (through the host: the other device's context is busy with bins of its own)
                                if (stagingBytes < part.bytes) { staging.reset(new uint8_t[part.bytes]); stagingBytes = part.bytes; }
                                GPU(isaac_gpu_download(part.tile->worker->ctx, staging.get(), part.block->as<uint8_t>() + part.offset, part.bytes));
                                GPU(isaac_gpu_upload(w.ctx, base, staging.get(), part.bytes));
                                part.block.reset();
                            }
                            else { GPU(isaac_gpu_upload(w.ctx, base, part.data.get(), part.bytes)); part.data.reset(); }
                            at += part.bytes;
                        }
                        isaac_bam_tile &b = bamTiles[i];
                        b.bcl_dev = base;
                        b.fragments_dev = reinterpret_cast<const isaac_fragment *>(base + align64(part.clusters * clusterLength));
                        b.cigar_dev = reinterpret_cast<const uint32_t *>(base + align64(align64(part.clusters * clusterLength) + part.clusters * nReads * sizeof(isaac_fragment)));
                        b.n_records = part.clusters * nReads;
                        b.read_name_prefix = part.tile->namePrefix.c_str(); b.read_group = part.tile->readGroup.c_str(); b.tls = &part.tile->tls;
                    }
                    lap(w.uploadSeconds);
                    isaac_bam_options options = bamOptions;
                    if (fileOrder[k] == nContigs) { options.bin_first_contig = 0; options.bin_end_contig = 0; options.bin_unaligned = 1; }
                    else { options.bin_first_contig = fileOrder[k]; options.bin_end_contig = fileOrder[k] + 1; options.bin_unaligned = 0; }
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
                        // BGZF on the device: stored blocks at level 0 (bgzf::BgzfCompressor's own), deflated ones otherwise
                        const uint64_t bound = o.bamGzipLevel ? isaac_gpu_bgzf_deflate_bound(nBytes) : isaac_gpu_bgzf_store_bound(nBytes);
                        if (bgzf.bytes() < bound) bgzf.reset(w.ctx, bound);
                        uint64_t nOut = 0;
                        if (o.bamGzipLevel) GPU(isaac_gpu_bgzf_deflate(w.ctx, bam.as<uint8_t>(), nBytes, 0, bgzf.as<uint8_t>(), bgzf.bytes(), &nOut));
                        else GPU(isaac_gpu_bgzf_store(w.ctx, bam.as<uint8_t>(), nBytes, 0, bgzf.as<uint8_t>(), bgzf.bytes(), &nOut));
                        lap(w.deflateSeconds);
                        result.bgzf.reset(new uint8_t[nOut]); result.bgzfBytes = nOut; result.recordsBytes = nBytes;
                        result.entries.reset(new isaac_bam_index_entry[result.nRecords]);
                        GPU(isaac_gpu_download(w.ctx, result.bgzf.get(), bgzf.as<uint8_t>(), nOut));
                        GPU(isaac_gpu_download(w.ctx, result.entries.get(), entries.as<isaac_bam_index_entry>(), result.nRecords * sizeof(isaac_bam_index_entry)));      // for the index
                        lap(w.downloadSeconds);
                    }
                    std::vector<BinPart>().swap(bin.parts);
                }
            }
            catch (const std::exception &e) { result.error = e.what(); }
            result.ready = true;
            { std::lock_guard<std::mutex> guard(outputLock); outputs[k] = std::move(result); }
            outputReady.notify_all();
        }
        w.buildSeconds = seconds() - start;
    };
    std::vector<std::thread> builders;
    for (auto &w : workers) builders.emplace_back(buildBins, std::ref(*w));

    const std::string directory = o.outputDirectory + "/Projects/default/default";
    makeDirectories(directory);
    const std::string bamPath = directory + "/sorted.bam";
    uint64_t nRecordsWritten = 0, binsWritten = 0;
    double writeSeconds = 0;
    std::string failure;
    {
        std::ofstream os(bamPath.c_str(), std::ios::binary | std::ios::trunc);
        if (!os) failure = "Failed to open output BAM file " + bamPath;
        os.write(reinterpret_cast<const char *>(headerBgzf.data()), std::streamsize(headerBgzf.size()));
        isaac_bam_indexer *indexer = isaac_gpu_bam_indexer_create(nContigs, headerBgzf.size());
        for (size_t k = 0; k < outputs.size(); ++k)
        {
            BinOutput out;
            {
                std::unique_lock<std::mutex> guard(outputLock);
                outputReady.wait(guard, [&] { return outputs[k].ready; });
                out = std::move(outputs[k]);
            }
            if (!out.error.empty() && failure.empty()) failure = out.error;
            if (!failure.empty() || !out.bgzfBytes) continue;
            const double writeStart = seconds();
            os.write(reinterpret_cast<const char *>(out.bgzf.get()), std::streamsize(out.bgzfBytes));
            writeSeconds += seconds() - writeStart;
            if (isaac_gpu_bam_indexer_add_entries(indexer, out.entries.get(), out.nRecords, out.recordsBytes, out.bgzf.get(), out.bgzfBytes)) failure = std::string("isaac_gpu_bam_indexer_add_entries: ") + isaac_gpu_bam_index_last_error();
            nRecordsWritten += out.nRecords; ++binsWritten;
        }
        for (std::thread &t : builders) t.join();
        os.write(reinterpret_cast<const char *>(eofBlock.data()), std::streamsize(eofBlock.size()));
        if (failure.empty() && !os) failure = "Failed to write " + bamPath;
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
    uint64_t tilesOnDevice = 0;
    for (auto &w : workers) tilesOnDevice += w->tilesKeptOnDevice;
    std::cerr << "isaac-align: " << bamPath << ": " << nRecordsWritten << " records in " << binsWritten << " bin(s)" << std::endl;
    // one line for scripts (bench.py): what the run took, stage by stage
    std::cerr << "isaac-align: timing {\"clusters\": " << totalClusters << ", \"reads\": " << totalClusters * nReads << ", \"records\": " << nRecordsWritten << ", \"workers\": " << workers.size()
              << ", \"reference_s\": " << referenceSeconds << ", \"reference_fasta_s\": " << fastaSeconds << ", \"reference_contigs_s\": " << contigSeconds << ", \"reference_table_s\": " << tableSeconds << ", \"load_and_find_s\": " << loadSeconds << ", \"load_text_wait_s\": " << g_textWaitSeconds << ", \"load_convert_s\": " << g_convertSeconds << ", \"load_first_lookup_s\": " << g_firstLookupSeconds << ", \"select_and_bin_s\": " << selectSeconds << ", \"build_and_write_s\": " << buildSeconds
              << ", \"tiles_kept_on_device\": " << tilesOnDevice << ", \"build_upload_s\": " << workers[0]->uploadSeconds << ", \"build_records_s\": " << workers[0]->recordsSeconds << ", \"build_deflate_s\": " << workers[0]->deflateSeconds
              << ", \"build_download_s\": " << workers[0]->downloadSeconds << ", \"file_write_s\": " << writeSeconds
              << ", \"total_s\": " << total << "}" << std::endl;
    return 0;
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
