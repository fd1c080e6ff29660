#include "align_options.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <thread>

namespace isaac_host
{
namespace
{

// one row of the option table: how the value is read and what is done with it
struct OptionSpec
{
    const char *name; char shortName; bool multitoken;
    std::function<void(const std::string &)> store;
    const char *help;
};

bool parseBool(const std::string &name, std::string v)
{   // boost::program_options: 1/0, true/false, yes/no, on/off, any case
    std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return char(std::tolower(c)); });
    if (v == "1" || v == "true" || v == "yes" || v == "on") return true;
    if (v == "0" || v == "false" || v == "no" || v == "off") return false;
    throw InvalidOption("the argument ('" + v + "') for option '--" + name + "' is invalid. Valid choices are 'on|off', 'yes|no', '1|0' and 'true|false'");
}
long parseNumber(const std::string &name, const std::string &v, bool allowNegative)
{
    char *end = 0;
    const long n = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || *end || (!allowNegative && (n < 0 || v[0] == '-'))) throw InvalidOption("the argument ('" + v + "') for option '--" + name + "' is invalid");
    return n;
}

} // namespace

std::string AlignOptions::usage()
{
    return
        "Usage: isaac-align -r <sorted-reference.xml> -b <fastq directory> --base-calls-format fastq|fastq-gz [options]\n"
        "\n"
        "Aligns the FASTQ lanes of the base calls directories on one MI355X and writes\n"
        "<output-directory>/Projects/default/default/sorted.bam and sorted.bam.bai.\n"
        "\n"
        "  -r [ --reference-genome ] arg        sorted-reference.xml of isaac-sort-reference (32-mer masks)\n"
        "  -b [ --base-calls ] arg              directory with lane<N>_read<R>.fastq[.gz]; one per flowcell\n"
        "  --base-calls-format arg              fastq | fastq-gz (bcl, bcl-gz and bam are not read by this host)\n"
        "  -o [ --output-directory ] arg (=./Aligned)\n"
        "  -t [ --temp-directory ] arg (=./Temp)   where the bins' parts go that neither the device nor the host's memory hold\n"
        "  -m [ --memory-limit ] arg (=0)       gigabytes of host memory the bins' parts may take before they go to files under -t\n"
        "                                       (0: no limit; the reference's limit of its whole process)\n"
        "  -j [ --jobs ] arg                    host threads for BGZF compression and FASTQ inflation\n"
        "  --device arg (=0)                    HIP device\n"
        "  --devices arg                        HIP devices, comma separated: one worker per entry; the tiles and the bins of the\n"
        "                                       run are dealt out to them (an entry may repeat: two workers on one device)\n"
        "  --bin-records arg (=0)               records a bin of the BAM stage is sized for (0: 4000000): contigs are grouped into or cut\n"
        "                                       into bins of about that many, each sorted, filtered and realigned by itself\n"
        "  --default-adapters arg               sequencing adapters to clip, one entry per base-calls directory (the flowcells of a run must agree):\n"
        "                                       Standard | Nextera | NexteraMp | a comma separated list of ACGT (any strand), ACGT* (forward\n"
        "                                       alignments, everything from the adapter on) and *ACGT (reverse alignments), in the direction of the reference\n"
        "  --use-bases-mask arg (=default)      y*n per read by default (the last cycle is not used); y<N>n<M> and y* forms\n"
        "  --seeds arg (=auto)                  auto | all | offsets 0:32:64[,...]\n"
        "  --first-pass-seeds arg (=1)\n"
        "  --repeat-threshold arg (=10)\n"
        "  --shadow-scan-range arg (=-1)\n"
        "  --clusters-at-a-time arg (=0)        clusters per load and per tile (0: 40000000 / seeds per tile)\n"
        "  --gap-scoring arg (=bwa)             bwa | eland | m:mm:go:ge:me\n"
        "  --gapped-mismatches arg (=5)\n"
        "  --semialigned-gap-limit arg (=100)\n"
        "  --base-quality-cutoff arg (=25)\n"
        "  --clip-semialigned arg (=1)\n"
        "  --clip-overlapping arg (=1)\n"
        "  --scatter-repeats arg (=0)\n"
        "  --ignore-neighbors arg (=0)\n"
        "  --mapq-threshold arg (=0)\n"
        "  --per-tile-tls arg (=0)\n"
        "  --dodgy-alignment-score arg (=0)     Unknown | Unaligned | 0-254\n"
        "  --keep-unaligned arg (=back)         discard | front | back\n"
        "  --realign-gaps arg (=sample)         no | sample | project | all (one sample: the last three are the same)\n"
        "  --realign-vigorously arg (=0)        try a realigned fragment again until nothing improves; fragments with up to thirty gaps in reach\n"
        "  --realign-dodgy arg (=0)\n"
        "  --realigned-gaps-per-fragment arg (=1)   taken and not needed (the reference sizes a reservation by it)\n"
        "  --mark-duplicates arg (=1)\n"
        "  --keep-duplicates arg (=1)\n"
        "  --bam-gzip-level arg (=1)\n"
        "  --bam-header-tag arg                 additional header lines, verbatim\n"
        "  --bam-pu-format arg (=%F:%L:%B)\n"
        "  --bam-pessimistic-mapq arg (=0)\n"
        "  --description arg                    @PG DS\n"
        "  --variable-read-length arg           reads shorter than the first one are padded with N\n"
        "  --lane-number-max arg (=8)\n"
        "  -h [ --help ], -v [ --version ]\n"
        "\n"
        "Environment switches (all off by default; none changes a record of the output):\n"
        "  ISAAC_ALIGN_STREAM_SELECTION=1|0     selections beside the loading (1) or after it (0), whatever the run's length\n"
        "  ISAAC_ALIGN_PLAN_ONLY=1              print the run's plan (reader threads, loads, bins) and stop before a device is touched\n"
        "  ISAAC_ALIGN_DUMP_TILES=<dir>:<lane>.<tile>,...   write the named tiles out as they were selected (for parity checks)\n"
        "  ISAAC_ALIGN_ORDERLY_EXIT=1           run every destructor at the end instead of leaving once the files are complete\n"
        "  ISAAC_ALIGN_NO_PREALLOCATION=1       no fallocate ahead of the BAM writes\n"
        "  ISAAC_ALIGN_READ_THREADS=<n>         threads that read lanes side by side (default: devices + 1, at most the lanes)\n"
        " the routes of a large run, forced on a small one (tests):\n"
        "  ISAAC_ALIGN_HOST_LOADS=1             base calls through host memory instead of staying on the device\n"
        "  ISAAC_ALIGN_HOST_BINS=1              bin parts in host memory; ISAAC_ALIGN_SPILL_BINS=1: in files under -t\n"
        "  ISAAC_ALIGN_STRANGERS=1              the workers of one device treat each other as other devices (with ISAAC_GPU_SHARE_BY_COPY=1)\n"
        " measurements:\n"
        "  ISAAC_ALIGN_BUILD_AHEAD=<n>          bins encoded ahead of the file writer\n"
        "  ISAAC_ALIGN_WARM_BUFFERS=1           touch the page-locked buffers before the clock starts\n"
        "  ISAAC_ALIGN_SYNC_DOWNLOADS=1         downloads on the work stream instead of the copy stream\n"
        "  ISAAC_ALIGN_TIMING_NO_RESOLUTION=1   skip isaac_gpu_resolve_flagged (timing only: MAPQs near an integer stay the device's)\n"
        "  ISAAC_ALIGN_MAPPED_WRITES=1          BAM written through a mapping instead of pwrite\n";
}

AlignOptions AlignOptions::parse(int argc, char **argv)
{
    AlignOptions o;
    o.argv.assign(argv, argv + argc);
    o.jobs = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::string> sampleSheet, referenceName, tiles, barcodeMismatches, useBasesMaskList;
    std::string startFrom = "Start", stopAt = "Finish", binRegex = "all", memoryControl, statsImageFormat;
    bool ignoreRepeats = false, avoidSmithWaterman = false, singleLibrarySamples = true, qscoreBin = false, pfOnly = true;
    unsigned neighborhoodSizeThreshold = 0;

    auto text = [](std::string *to) { return [to](const std::string &v) { *to = v; }; };
    auto list = [](std::vector<std::string> *to) { return [to](const std::string &v) { to->push_back(v); }; };
    auto ignored = [](const std::string &) {};
    std::vector<OptionSpec> specs;
    auto number = [&specs](const char *name, char shortName, unsigned *to) { specs.push_back({ name, shortName, false, [name, to](const std::string &v) { *to = unsigned(parseNumber(name, v, false)); }, "" }); };
    auto integer = [&specs](const char *name, int *to) { specs.push_back({ name, 0, false, [name, to](const std::string &v) { *to = int(parseNumber(name, v, true)); }, "" }); };
    auto flag = [&specs](const char *name, bool *to) { specs.push_back({ name, 0, false, [name, to](const std::string &v) { *to = parseBool(name, v); }, "" }); };
    specs.push_back({ "base-calls", 'b', true, list(&o.baseCalls), "" });
    specs.push_back({ "base-calls-directory", 0, true, list(&o.baseCalls), "" });
    specs.push_back({ "base-calls-format", 0, true, list(&o.baseCallsFormat), "" });
    specs.push_back({ "reference-genome", 'r', false, text(&o.referenceGenome), "" });
    specs.push_back({ "reference-name", 'n', true, list(&referenceName), "" });
    specs.push_back({ "output-directory", 'o', false, text(&o.outputDirectory), "" });
    specs.push_back({ "temp-directory", 't', false, text(&o.tempDirectory), "" });
    specs.push_back({ "seeds", 0, false, text(&o.seeds), "" });
    specs.push_back({ "gap-scoring", 0, false, text(&o.gapScoring), "" });
    specs.push_back({ "dodgy-alignment-score", 0, false, text(&o.dodgyAlignmentScore), "" });
    specs.push_back({ "keep-unaligned", 0, false, text(&o.keepUnaligned), "" });
    specs.push_back({ "realign-gaps", 0, false, text(&o.realignGaps), "" });
    specs.push_back({ "use-bases-mask", 0, true, list(&useBasesMaskList), "" });
    specs.push_back({ "bam-pu-format", 0, false, text(&o.bamPuFormat), "" });
    specs.push_back({ "bam-header-tag", 0, true, list(&o.bamHeaderTags), "" });
    specs.push_back({ "bam-exclude-tags", 0, false, text(&o.bamExcludeTags), "" });
    specs.push_back({ "description", 0, false, text(&o.description), "" });
    specs.push_back({ "tls", 0, false, text(&o.tls), "" });
    specs.push_back({ "devices", 0, false, text(&o.devices), "" });
    specs.push_back({ "sample-sheet", 's', true, list(&sampleSheet), "" });
    specs.push_back({ "tiles", 0, true, list(&tiles), "" });
    specs.push_back({ "default-adapters", 0, true, list(&o.defaultAdapters), "" });
    specs.push_back({ "barcode-mismatches", 0, true, list(&barcodeMismatches), "" });
    specs.push_back({ "start-from", 0, false, text(&startFrom), "" });
    specs.push_back({ "stop-at", 0, false, text(&stopAt), "" });
    specs.push_back({ "bin-regex", 0, false, text(&binRegex), "" });
    number("seed-length", 0, &o.seedLength); number("first-pass-seeds", 0, &o.firstPassSeeds); number("jobs", 'j', &o.jobs); number("repeat-threshold", 0, &o.repeatThreshold);
    number("lane-number-max", 0, &o.laneNumberMax); number("clusters-at-a-time", 0, &o.clustersAtATime); number("mapq-threshold", 0, &o.mapqThreshold);
    number("base-quality-cutoff", 0, &o.baseQualityCutoff); number("semialigned-gap-limit", 0, &o.semialignedGapLimit); number("gapped-mismatches", 0, &o.gappedMismatches);
    number("realigned-gaps-per-fragment", 0, &o.realignedGapsPerFragment); number("neighborhood-size-threshold", 0, &neighborhoodSizeThreshold);
    number("bin-records", 0, &o.binRecords);
    integer("shadow-scan-range", &o.shadowScanRange); integer("bam-gzip-level", &o.bamGzipLevel); integer("device", &o.device);
    flag("ignore-neighbors", &o.ignoreNeighbors); flag("per-tile-tls", &o.perTileTls); flag("scatter-repeats", &o.scatterRepeats); flag("clip-semialigned", &o.clipSemialigned);
    flag("clip-overlapping", &o.clipOverlapping); flag("realign-vigorously", &o.realignVigorously); flag("realign-dodgy", &o.realignDodgy); flag("keep-duplicates", &o.keepDuplicates);
    flag("mark-duplicates", &o.markDuplicates); flag("bam-pessimistic-mapq", &o.pessimisticMapQ); flag("variable-read-length", &o.variableReadLength);
    flag("variable-fastq-read-length", &o.variableFastqReadLength); flag("allow-empty-flowcells", &o.allowEmptyFlowcells); flag("ignore-repeats", &ignoreRepeats);
    flag("avoid-smith-waterman", &avoidSmithWaterman); flag("single-library-samples", &singleLibrarySamples); flag("qscore-bin", &qscoreBin); flag("pf-only", &pfOnly);
    // the reference's own resources: no effect here
    for (const char *name : { "input-parallel-load", "temp-parallel-load", "temp-parallel-save", "output-parallel-save", "verbosity", "memory-control", "cleanup-intermediary",
                              "expected-bgzf-ratio", "pre-sort-bins", "buffer-bins", "stats-image-format", "ignore-missing-bcls", "ignore-missing-filters" })
        specs.push_back({ name, 0, false, ignored, "" });
    number("memory-limit", 'm', &o.memoryLimit);

    auto find = [&specs](const std::string &name, bool isShort) -> const OptionSpec *
    {
        for (const OptionSpec &s : specs) if (isShort ? (s.shortName && name.size() == 1 && s.shortName == name[0]) : (name == s.name)) return &s;
        return 0;
    };
    for (int i = 1; i < argc; ++i)
    {
        const std::string arg = argv[i];
        if (arg == "-h" || arg == "--help") { o.action = HELP; return o; }
        if (arg == "-v" || arg == "--version") { o.action = VERSION; return o; }
        if (arg.size() < 2 || arg[0] != '-') throw InvalidOption("too many positional options have been specified on the command line");
        const bool isLong = arg[1] == '-';
        std::string name = isLong ? arg.substr(2) : arg.substr(1, 1), value;
        bool haveValue = false;
        if (isLong) { const size_t eq = name.find('='); if (std::string::npos != eq) { value = name.substr(eq + 1); name = name.substr(0, eq); haveValue = true; } }
        else if (arg.size() > 2) { value = arg.substr(2); haveValue = true; }
        const OptionSpec *spec = find(name, !isLong);
        if (!spec) throw InvalidOption("unrecognised option '" + arg + "'");
        if (haveValue) { spec->store(value); continue; }
        if (i + 1 >= argc) throw InvalidOption("the required argument for option '--" + std::string(spec->name) + "' is missing");
        spec->store(argv[++i]);
        // multitoken: everything up to the next option
        while (spec->multitoken && i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] && !std::isdigit(static_cast<unsigned char>(argv[i + 1][1])))) spec->store(argv[++i]);
    }

    // ---- AlignOptions::postProcess (AlignOptions.cpp:477-1310), the parts that apply
    if (o.baseCalls.empty()) throw InvalidOption("\n   *** At least one 'base-calls' is required ***\n");
    if (o.referenceGenome.empty()) throw InvalidOption("\n   *** At least one 'reference-genome' is required ***\n");
    if (o.baseCallsFormat.empty()) o.baseCallsFormat.push_back("bcl");
    if (o.baseCallsFormat.size() > o.baseCalls.size()) throw InvalidOption("\n   *** Too many --base-calls-format options specified. There must be at most one per --base-calls. ***\n");
    o.baseCallsFormat.resize(o.baseCalls.size(), o.baseCallsFormat.back());
    for (const std::string &format : o.baseCallsFormat)
    {
        if (format == "bcl" || format == "bcl-gz" || format == "bam") throw InvalidOption("\n   *** --base-calls-format " + format + ": this host reads fastq and fastq-gz only ***\n");
        if (format != "fastq" && format != "fastq-gz") throw InvalidOption("\n   *** --base-calls-format " + format + " is not supported ***\n");
    }
    if (useBasesMaskList.size() > o.baseCalls.size()) throw InvalidOption("\n   *** Too many --use-bases-mask options specified. There must be at most one per --base-calls. ***\n");
    if (!useBasesMaskList.empty())
    {
        for (const std::string &m : useBasesMaskList) if (m != useBasesMaskList.front()) throw InvalidOption("\n   *** different --use-bases-mask values per flowcell are not supported by this host ***\n");
        o.useBasesMask = useBasesMaskList.front();
    }
    if (16 != o.seedLength && 32 != o.seedLength && 64 != o.seedLength) throw InvalidOption("\n   *** --seed-length other than 16, 32 or 64 is not supported. ***\n");
    if (32 != o.seedLength) throw InvalidOption("\n   *** --seed-length " + std::to_string(o.seedLength) + ": the GPU path implements 32-mer seeds only ***\n");
    if (o.realignGaps == "yes") o.realignGaps = "sample";
    if (o.realignGaps != "no" && o.realignGaps != "sample" && o.realignGaps != "project" && o.realignGaps != "all")
        throw InvalidOption("\n   *** The 'realign-gaps' value is invalid " + o.realignGaps + " ***\n");
    if (o.keepUnaligned != "discard" && o.keepUnaligned != "front" && o.keepUnaligned != "back") throw InvalidOption("\n   *** The 'keep-unaligned' string must must be 'discard', 'front' or 'back'***\n");
    if (o.dodgyAlignmentScore != "Unknown" && o.dodgyAlignmentScore != "Unaligned")
    {
        char *end = 0;
        const long v = std::strtol(o.dodgyAlignmentScore.c_str(), &end, 10);
        if (o.dodgyAlignmentScore.empty() || *end || v < 0 || v > 254)
            throw InvalidOption("\n   *** The 'dodgy-alignment-score' option must be either Unknown, Unaligned or a number 0-255 (" + o.dodgyAlignmentScore + " given) ***\n");
    }
    if (o.bamGzipLevel < 0 || o.bamGzipLevel > 9) throw InvalidOption("\n   *** --bam-gzip-level must be between 0 and 9 ***\n");
    if (!o.firstPassSeeds) throw InvalidOption("\n   *** At least one seed must be used on the first pass (--first-pass-seeds is 0) ***\n");
    // what the GPU path does not do: refused rather than silently different
    auto refuse = [](bool condition, const std::string &what) { if (condition) throw InvalidOption("\n   *** " + what + " is not supported by this host ***\n"); };
    refuse(o.bamExcludeTags != "ZX,ZY", "--bam-exclude-tags other than ZX,ZY");
    refuse(!o.tls.empty(), "--tls");
    // --realigned-gaps-per-fragment is "an estimate of how many gaps the realignment will introduce into each fragment" (AlignOptions.cpp:428-429): in the reference it
    // only sizes a reservation (GapRealigner.hh:208-209); the realigner here sizes its CIGAR pool from the bin itself, so any value is taken and none changes a record
    refuse(ignoreRepeats, "--ignore-repeats 1");
    refuse(avoidSmithWaterman, "--avoid-smith-waterman 1");
    refuse(qscoreBin, "--qscore-bin 1");
    refuse(0 != neighborhoodSizeThreshold, "--neighborhood-size-threshold other than 0");
    refuse(!singleLibrarySamples, "--single-library-samples 0");
    refuse(startFrom != "Start" || stopAt != "Finish", "--start-from / --stop-at");
    refuse(binRegex != "all", "--bin-regex other than all");
    for (const std::string &s : sampleSheet) refuse(s != "none", "--sample-sheet (other than none)");
    for (const std::string &s : referenceName) refuse(s != "default", "--reference-name other than default");
    for (const std::string &s : tiles) refuse(!s.empty(), "--tiles");
    // --default-adapters: entry i belongs to base-calls i, a flowcell without one gets nothing clipped (AlignOptions.cpp:189-207,1248-1249).  The run has one set of
    // parameters on the device, so the flowcells of a run have to agree; each entry is parsed here so that a bad one stops the run before anything is loaded.
    {
        isaac_params probe;
        std::memset(&probe, 0, sizeof(probe));
        for (const std::string &s : o.defaultAdapters) if (isaac_gpu_parse_adapters(s.c_str(), &probe)) throw InvalidOption(isaac_gpu_params_last_error());
        const std::string first = o.defaultAdapters.empty() ? std::string() : o.defaultAdapters[0];
        for (size_t i = 1; i < std::max(o.baseCalls.size(), o.defaultAdapters.size()); ++i)
            refuse((i < o.defaultAdapters.size() ? o.defaultAdapters[i] : std::string()) != first, "flowcells with different --default-adapters in one run");
    }
    (void)pfOnly;       // FASTQ data has no filter files: every cluster passes
    return o;
}

std::vector<int> AlignOptions::deviceList() const
{
    std::vector<int> list;
    if (devices.empty()) { list.push_back(device); return list; }
    for (size_t at = 0; at <= devices.size();)
    {
        const size_t comma = std::min(devices.find(',', at), devices.size());
        const std::string item = devices.substr(at, comma - at);
        char *end = 0;
        const long v = std::strtol(item.c_str(), &end, 10);
        if (item.empty() || *end || v < 0) throw InvalidOption("\n   *** --devices: a comma separated list of device numbers is expected (" + devices + " given) ***\n");
        list.push_back(int(v));
        at = comma + 1;
    }
    return list;
}

unsigned AlignOptions::forcedDodgyAlignmentScore() const
{   // AlignWorkflow.cpp:391-392
    if ("Unknown" == dodgyAlignmentScore) return 255;
    if ("Unaligned" == dodgyAlignmentScore) return 0;
    return unsigned(std::atoi(dodgyAlignmentScore.c_str()));
}

isaac_params AlignOptions::params(unsigned readLength1, unsigned readLength2) const
{
    isaac_params p;
    if (isaac_gpu_default_params(readLength1, readLength2, &p)) throw InvalidOption(isaac_gpu_params_last_error());
    p.repeat_threshold = repeatThreshold; p.gapped_mismatches_max = gappedMismatches; p.semialigned_gap_limit = semialignedGapLimit; p.base_quality_cutoff = baseQualityCutoff;
    p.ignore_neighbors = ignoreNeighbors; p.clip_semialigned = clipSemialigned; p.clip_overlapping = clipOverlapping; p.scatter_repeats = scatterRepeats;
    p.dodgy_alignment_score = "Unknown" == dodgyAlignmentScore ? 255 : "Unaligned" == dodgyAlignmentScore ? -1 : std::atoi(dodgyAlignmentScore.c_str());
    p.mapq_threshold = mapqThreshold; p.keep_unaligned = keepUnalignedRecords(); p.mate_drift_range = shadowScanRange; p.seed_length = seedLength;
    if (isaac_gpu_parse_gap_scoring(gapScoring.c_str(), &p) || isaac_gpu_parse_seeds(seeds.c_str(), firstPassSeeds, &p)) throw InvalidOption(isaac_gpu_params_last_error());
    if (!defaultAdapters.empty() && isaac_gpu_parse_adapters(defaultAdapters[0].c_str(), &p)) throw InvalidOption(isaac_gpu_params_last_error());
    return p;
}

} // namespace isaac_host
