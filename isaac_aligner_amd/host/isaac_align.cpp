// isaac-align on one MI355X: the host the reference's bin/isaac-align.cpp + workflow::AlignWorkflow are for FASTQ data and one sample,
// written against include/isaac_gpu.h only.
//   options        options::AlignOptions                                        align_options.cpp
//   flowcells      options::alignOptions::FastqFlowcell, FastqSeedSource        fastq_flowcell.cpp; tiles by isaac_gpu_fastq_tiles
//   reference      reference::loadSortedReferenceXml, reference::loadContigs    (lib/reference/ContigLoader.cpp:29-66) + isaac_gpu_load_sorted_reference
//   find matches   workflow::alignWorkflow::FindMatchesTransition               isaac_gpu_fastq_to_bcl, isaac_gpu_find_matches (which contigs have matches)
//   select         workflow::alignWorkflow::SelectMatchesTransition             isaac_gpu_find_matches again (the matches are not kept: 1.3 ms per million
//                                                                               pairs against 400 bytes per pair), isaac_gpu_determine_tls per lane
//                                                                               (MatchSelector.cpp:395-412), isaac_gpu_select, isaac_gpu_compact_cigars
//   build          build::Build                                                 isaac_gpu_bam_records over all tiles, one BGZF run per contig, sorted.bam + .bai
// Everything a tile needs later stays in HBM: BCL bytes (the BAM records are made from them), 64-byte records, packed CIGARs.
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
    std::string bases; std::vector<uint64_t> offsets;
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
    // reference::loadContig: the alphabetic characters from the contig's offset on, ACGT as they are (upper case), everything else N
    ref.offsets.push_back(0);
    for (const isaac_reference_contig &c : ref.contigs)
    {
        std::string path = c.file;
        if (!fileExists(path) && '/' != path[0] && fileExists(directoryOf(xmlPath) + "/" + path)) path = directoryOf(xmlPath) + "/" + path;
        std::ifstream fasta(path.c_str(), std::ios::binary);
        if (!fasta) throw std::runtime_error("Failed to open reference file " + path);
        if (!fasta.seekg(std::streamoff(c.offset))) throw std::runtime_error("Failed to reach offset " + std::to_string(c.offset) + " in reference file " + path);
        const size_t before = ref.bases.size();
        ref.bases.reserve(before + c.total_bases);
        std::vector<char> buffer(1 << 20);
        while (ref.bases.size() - before < c.total_bases && fasta)
        {
            fasta.read(buffer.data(), std::streamsize(buffer.size()));
            const std::streamsize got = fasta.gcount();
            for (std::streamsize i = 0; i < got && ref.bases.size() - before < c.total_bases; ++i)
            {
                const unsigned char b = static_cast<unsigned char>(buffer[size_t(i)]);
                if (!std::isalpha(b)) continue;
                const char u = char(std::toupper(b));
                ref.bases.push_back(('A' == u || 'C' == u || 'G' == u || 'T' == u) ? u : 'N');
            }
        }
        if (ref.bases.size() - before != c.total_bases)
            throw std::runtime_error("Failed to read " + std::to_string(c.total_bases) + " bases from reference file " + path + ": " + std::to_string(ref.bases.size() - before));
        ref.offsets.push_back(ref.bases.size());
    }
    return ref;
}

// ---- the data: lanes, loads, tiles -------------------------------------------------------------------------------------------------------
struct Tile
{
    unsigned lane = 0, number = 0, index = 0, clusters = 0;     // tile number within the lane (the read name), index over the run (the records)
    const uint8_t *bcl = 0;                                       // inside its load's buffer
    DeviceMemory records, cigars;
    isaac_tls tls;
    std::string namePrefix, readGroup;
};

// one read of one lane: the file and the text not yet converted
struct ReadStream
{
    std::unique_ptr<FastqFileReader> reader;
    std::vector<char> pending;
    uint64_t consumedBytes = 0;                                   // of the uncompressed text, for error messages
};

const size_t TEXT_CHUNK = size_t(64) << 20;

// io::FastqLoader::loadSingleRead for up to maxClusters clusters: the text goes to the device in pieces, the converter leaves the incomplete
// record at the end of a piece for the next one
uint32_t loadRead(isaac_gpu_ctx *ctx, ReadStream &stream, unsigned readIndex, bool allowVariableLength, uint8_t *bclDev, unsigned clusterLength, uint32_t maxClusters, DeviceMemory &textDev)
{
    uint32_t clusters = 0;
    size_t chunk = TEXT_CHUNK;
    while (clusters < maxClusters)
    {
        if (stream.pending.size() < chunk && !stream.reader->atEnd()) stream.reader->read(stream.pending, chunk - stream.pending.size());
        if (stream.pending.empty()) break;
        const bool final = stream.reader->atEnd();
        if (textDev.bytes() < stream.pending.size() + 64) textDev.reset(ctx, stream.pending.size() + 64);
        GPU(isaac_gpu_upload(ctx, textDev.as<char>(), stream.pending.data(), stream.pending.size()));
        uint32_t n = 0; uint64_t consumed = 0, errorOffset = 0;
        const int rc = isaac_gpu_fastq_to_bcl(ctx, textDev.as<char>(), stream.pending.size(), readIndex, allowVariableLength, final, bclDev + uint64_t(clusters) * clusterLength,
                                              maxClusters - clusters, &n, &consumed, &errorOffset);
        if (rc)
            throw std::runtime_error(stream.reader->path() + ": " + isaac_gpu_last_error() + " (record " + std::to_string(clusters + n) + " of this load, offset " +
                                     std::to_string(stream.consumedBytes + errorOffset) + ")");
        clusters += n;
        stream.consumedBytes += consumed;
        stream.pending.erase(stream.pending.begin(), stream.pending.begin() + std::ptrdiff_t(consumed));
        if (!n && !consumed)
        {
            if (final) break;                                     // nothing but line ends left
            chunk *= 2;                                           // a record longer than the piece
        }
    }
    return clusters;
}

struct Part { uint64_t offset, bytes; std::vector<uint8_t> bgzf; };

int run(const AlignOptions &o)
{
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

    // ---- reference
    isaac_gpu_ctx *ctx = 0;
    GPU(isaac_gpu_create(o.device, &params, 0, &ctx));
    std::unique_ptr<isaac_gpu_ctx, void (*)(isaac_gpu_ctx *)> ctxGuard(ctx, isaac_gpu_destroy);
    Reference reference;
    {
        Stage stage("loading the reference");
        reference = loadReference(o.referenceGenome);
        GPU(isaac_gpu_load_contigs(ctx, reference.bases.data(), reference.offsets.data(), uint32_t(reference.contigs.size())));
        std::string().swap(reference.bases);
        GPU(isaac_gpu_load_sorted_reference(ctx, o.referenceGenome.c_str()));
    }
    const uint32_t nContigs = uint32_t(reference.contigs.size());

    // ---- FastqSeedSource: loads of --clusters-at-a-time clusters, tiles of at most tileClustersMax
    const uint32_t tileClustersMax = isaac_gpu_fastq_tile_clusters_max(o.clustersAtATime, params.n_seeds);
    const uint32_t loadClusters = o.clustersAtATime ? o.clustersAtATime : 4 * tileClustersMax;      // a multiple of the tile size: the tiles come out the same for any such load
    std::deque<Tile> tiles;
    std::vector<DeviceMemory> loads;
    std::vector<uint8_t> contigHasMatches(nContigs, 0);
    DeviceMemory matches, offsets(ctx, (uint64_t(tileClustersMax) + 1) * 8), textDev;
    uint64_t matchCapacity = 0;
    auto findMatches = [&](const Tile &t, uint64_t &nMatches)
    {
        const uint64_t worst = uint64_t(t.clusters) * 2 * params.n_seeds * std::max(1u, params.repeat_threshold - 1);
        if (!matchCapacity) { matchCapacity = std::max<uint64_t>(1024, std::min<uint64_t>(worst, uint64_t(tileClustersMax) * 24)); matches.reset(ctx, matchCapacity * sizeof(isaac_match)); }
        for (;;)
        {
            const int rc = isaac_gpu_find_matches(ctx, t.bcl, t.clusters, t.index, matches.as<isaac_match>(), matchCapacity, offsets.as<uint64_t>(), &nMatches, contigHasMatches.data());
            if (ISAAC_GPU_ECAPACITY != rc) { check(rc, "isaac_gpu_find_matches"); return; }
            matchCapacity = std::max(nMatches, 2 * matchCapacity);
            matches.reset(ctx, matchCapacity * sizeof(isaac_match));
        }
    };
    {
        Stage stage("loading base calls and finding matches");
        unsigned barcodeIndex = 0;
        uint64_t totalClusters = 0;
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
                    DeviceMemory bcl(ctx, uint64_t(loadClusters) * clusterLength + 64);
                    uint32_t loaded[2] = { 0, 0 };
                    for (unsigned r = 0; r < nReads; ++r)
                        loaded[r] = loadRead(ctx, streams[r], r, o.variableReadLength || o.variableFastqReadLength, bcl.as<uint8_t>(), clusterLength, loadClusters, textDev);
                    if (2 == nReads && loaded[0] != loaded[1])
                        throw std::runtime_error("Mismatching number of clusters in " + lane.readPath[0] + " (" + std::to_string(loaded[0]) + ") and " + lane.readPath[1] + " (" + std::to_string(loaded[1]) + ")");
                    if (!loaded[0]) break;
                    if (loaded[0] < loadClusters / 2)
                    {   // "allocated too much memory for bcl data": the load keeps what it uses
                        DeviceMemory exact(ctx, uint64_t(loaded[0]) * clusterLength + 64);
                        GPU(isaac_gpu_copy(ctx, exact.as<uint8_t>(), bcl.as<uint8_t>(), uint64_t(loaded[0]) * clusterLength));
                        GPU(isaac_gpu_synchronize(ctx));
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
                        t.lane = lane.lane; t.number = numbers[k]; t.index = unsigned(tiles.size() - 1); t.clusters = sizes[k]; t.bcl = bcl.as<uint8_t>() + first * clusterLength;
                        t.namePrefix = fc.flowcellId + ":" + std::to_string(lane.lane) + ":" + std::to_string(t.number) + ":"; t.readGroup = readGroup;
                        std::memset(&t.tls, 0, sizeof(t.tls));
                        first += sizes[k];
                        uint64_t nMatches = 0;
                        findMatches(t, nMatches);
                    }
                    totalClusters += loaded[0];
                    loads.push_back(std::move(bcl));
                    if (loaded[0] < loadClusters) break;
                }
            }
        textDev.release();
        std::cerr << "isaac-align: " << totalClusters << " clusters in " << tiles.size() << " tile(s)" << std::endl;
        if (tiles.empty()) throw InvalidOption("No data found to process. Please check your --base-calls.");
    }

    // ---- SelectMatchesTransition: every tile with the contigs the whole run has matches on
    {
        Stage stage("selecting matches");
        GPU(isaac_gpu_set_loaded_contigs(ctx, contigHasMatches.data(), nContigs));
        DeviceMemory slots(ctx, uint64_t(tileClustersMax) * nReads * ISAAC_GPU_MAX_CIGAR_OPS * 4), packed;
        isaac_tls tls; std::memset(&tls, 0, sizeof(tls));
        std::string laneKey;
        for (Tile &t : tiles)
        {
            uint64_t nMatches = 0;
            findMatches(t, nMatches);
            if (laneKey != t.readGroup) { std::memset(&tls, 0, sizeof(tls)); laneKey = t.readGroup; }       // barcodeTemplateLengthStatistics: one per barcode
            if (!tls.stable || o.perTileTls)
            {
                GPU(isaac_gpu_determine_tls(ctx, t.bcl, t.clusters, t.index, matches.as<isaac_match>(), offsets.as<uint64_t>(), &tls));
                std::cerr << "isaac-align: template length statistics of tile " << t.namePrefix << " min " << tls.min << " median " << tls.median << " max " << tls.max
                          << (tls.stable ? " (stable)" : " (unstable)") << std::endl;
            }
            t.tls = tls;
            const uint64_t nRecords = uint64_t(t.clusters) * nReads;
            t.records.reset(ctx, nRecords * sizeof(isaac_fragment));
            GPU(isaac_gpu_select(ctx, t.bcl, t.clusters, t.index, matches.as<isaac_match>(), offsets.as<uint64_t>(), &tls, t.records.as<isaac_fragment>(), slots.as<uint32_t>(),
                                 nRecords * ISAAC_GPU_MAX_CIGAR_OPS));
            // the CIGARs as the bin files hold them: back to back
            uint64_t words = 0;
            if (packed.bytes() < nRecords * 8 * 4) packed.reset(ctx, nRecords * 8 * 4);
            int rc = isaac_gpu_compact_cigars(ctx, t.records.as<isaac_fragment>(), nRecords, slots.as<uint32_t>(), packed.as<uint32_t>(), packed.bytes() / 4, &words);
            if (ISAAC_GPU_ECAPACITY == rc)
            {
                packed.reset(ctx, words * 4);
                rc = isaac_gpu_compact_cigars(ctx, t.records.as<isaac_fragment>(), nRecords, slots.as<uint32_t>(), packed.as<uint32_t>(), packed.bytes() / 4, &words);
            }
            check(rc, "isaac_gpu_compact_cigars");
            t.cigars.reset(ctx, words * 4);
            GPU(isaac_gpu_copy(ctx, t.cigars.as<uint32_t>(), packed.as<uint32_t>(), words * 4));
        }
        GPU(isaac_gpu_synchronize(ctx));
        isaac_counters counters;
        if (!isaac_gpu_get_counters(ctx, &counters) && counters.overflow_clusters)
            std::cerr << "WARNING: " << counters.overflow_clusters << " cluster(s) of the last tile exceeded a fixed work list; their records are flagged (isaac_fragment::reserved bit 2)" << std::endl;
    }
    matches.release(); offsets.release();

    // ---- build::Build: the record stream of the whole run, then one BGZF run per bin (contig) and the index over them
    std::vector<uint8_t> stream;
    uint64_t nRecordsWritten = 0, unalignedOffset = 0;
    {
        Stage stage("making BAM records");
        std::vector<isaac_bam_tile> bamTiles(tiles.size());
        uint64_t nRecords = 0;
        for (size_t i = 0; i < tiles.size(); ++i)
        {
            const Tile &t = tiles[i];
            isaac_bam_tile &b = bamTiles[i];
            b.bcl_dev = t.bcl; b.fragments_dev = t.records.as<isaac_fragment>(); b.cigar_dev = t.cigars.as<uint32_t>(); b.n_records = uint64_t(t.clusters) * nReads;
            b.read_name_prefix = t.namePrefix.c_str(); b.read_group = t.readGroup.c_str(); b.tls = &t.tls;
            nRecords += b.n_records;
        }
        isaac_bam_options bamOptions; std::memset(&bamOptions, 0, sizeof(bamOptions));
        bamOptions.forced_dodgy_alignment_score = o.forcedDodgyAlignmentScore(); bamOptions.pessimistic_mapq = o.pessimisticMapQ; bamOptions.read_group = "0"; bamOptions.barcode = "none";
        bamOptions.mark_duplicates = o.markDuplicates; bamOptions.keep_duplicates = o.keepDuplicates; bamOptions.realign_gaps = "no" != o.realignGaps; bamOptions.realign_dodgy = o.realignDodgy;
        uint64_t capacity = nRecords * (96 + 2 * std::max(params.read_length[0], params.read_length[1])), nBytes = 0;
        DeviceMemory bam(ctx, capacity);
        int rc = isaac_gpu_bam_records(ctx, bamTiles.data(), uint32_t(bamTiles.size()), &bamOptions, bam.as<uint8_t>(), capacity, &nBytes, &nRecordsWritten, &unalignedOffset);
        if (ISAAC_GPU_ECAPACITY == rc)
        {
            capacity = nBytes; bam.reset(ctx, capacity);
            rc = isaac_gpu_bam_records(ctx, bamTiles.data(), uint32_t(bamTiles.size()), &bamOptions, bam.as<uint8_t>(), capacity, &nBytes, &nRecordsWritten, &unalignedOffset);
        }
        check(rc, "isaac_gpu_bam_records");
        stream.resize(nBytes);
        if (nBytes) GPU(isaac_gpu_download(ctx, stream.data(), bam.as<uint8_t>(), nBytes));
        std::cerr << "isaac-align: " << nRecordsWritten << " records, " << nBytes << " bytes" << std::endl;
    }

    Stage stage("writing sorted.bam");
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
    const auto compress = [&o](const uint8_t *data, uint64_t n, int eofBlock)
    {
        std::vector<uint8_t> out(isaac_gpu_bgzf_bound(n) + 64);
        uint64_t nOut = 0;
        if (isaac_gpu_bgzf_compress(data, n, o.bamGzipLevel, o.jobs, eofBlock, out.data(), out.size(), &nOut)) throw std::runtime_error(std::string("isaac_gpu_bgzf_compress: ") + isaac_gpu_bam_last_error());
        out.resize(nOut);
        return out;
    };
    const std::vector<uint8_t> headerBgzf = compress(header.data(), header.size(), 0);
    // the bins: every contig's records, and the unaligned ones
    std::vector<Part> parts;
    for (uint64_t at = 0; at < unalignedOffset; )
    {
        const auto le32 = [](const uint8_t *p) { return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24; };
        const uint32_t contig = le32(stream.data() + at + 4);
        const uint64_t begin = at;
        while (at < unalignedOffset && le32(stream.data() + at + 4) == contig) at += uint64_t(le32(stream.data() + at)) + 4;
        parts.push_back(Part{ begin, at - begin, {} });
    }
    const size_t alignedParts = parts.size();
    if (unalignedOffset < stream.size()) parts.push_back(Part{ unalignedOffset, stream.size() - unalignedOffset, {} });
    if ("front" == o.keepUnaligned && parts.size() > alignedParts) std::rotate(parts.begin(), parts.begin() + std::ptrdiff_t(alignedParts), parts.end());    // --keep-unaligned front
    for (Part &p : parts) p.bgzf = compress(stream.data() + p.offset, p.bytes, 0);
    const std::vector<uint8_t> eofBlock = compress(0, 0, 1);

    const std::string directory = o.outputDirectory + "/Projects/default/default";
    makeDirectories(directory);
    const std::string bamPath = directory + "/sorted.bam";
    {
        std::ofstream os(bamPath.c_str(), std::ios::binary | std::ios::trunc);
        if (!os) throw std::runtime_error("Failed to open output BAM file " + bamPath);
        os.write(reinterpret_cast<const char *>(headerBgzf.data()), std::streamsize(headerBgzf.size()));
        for (const Part &p : parts) os.write(reinterpret_cast<const char *>(p.bgzf.data()), std::streamsize(p.bgzf.size()));
        os.write(reinterpret_cast<const char *>(eofBlock.data()), std::streamsize(eofBlock.size()));
        if (!os) throw std::runtime_error("Failed to write " + bamPath);
    }
    std::vector<isaac_bam_index_part> indexParts;
    for (const Part &p : parts) indexParts.push_back(isaac_bam_index_part{ p.offset, p.bytes, p.bgzf.data(), p.bgzf.size() });
    uint64_t baiBytes = 0;
    isaac_gpu_bam_index(stream.data(), indexParts.data(), uint32_t(indexParts.size()), nContigs, headerBgzf.size(), 0, 0, &baiBytes);
    std::vector<uint8_t> bai(baiBytes);
    if (isaac_gpu_bam_index(stream.data(), indexParts.data(), uint32_t(indexParts.size()), nContigs, headerBgzf.size(), bai.data(), bai.size(), &baiBytes))
        throw std::runtime_error(std::string("isaac_gpu_bam_index: ") + isaac_gpu_bam_index_last_error());
    {
        std::ofstream os((bamPath + ".bai").c_str(), std::ios::binary | std::ios::trunc);
        if (!os || !os.write(reinterpret_cast<const char *>(bai.data()), std::streamsize(bai.size()))) throw std::runtime_error("Error opening bam index file for writing " + bamPath + ".bai");
    }
    std::cerr << "isaac-align: " << bamPath << ": " << nRecordsWritten << " records in " << parts.size() << " bin(s)" << std::endl;
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
