// The command line of isaac-align as options::AlignOptions defines it (lib/options/AlignOptions.cpp:77-475 for names and defaults,
// :477-1310 for the checks), reduced to what this host can honour: FASTQ input, one reference, no sample sheet.  Options that steer the
// reference's own resources (threads per stage, memory limits, temp files) are accepted and have no effect; options that would change
// the result in a way the GPU path does not implement are refused with a message instead of being ignored.
#pragma once
#include "isaac_gpu.h"

#include <stdexcept>
#include <string>
#include <vector>

namespace isaac_host
{

// common::InvalidOptionException: the program ends with the message and exit code 1 (include/common/Program.hh:60-92)
struct InvalidOption : std::runtime_error { explicit InvalidOption(const std::string &what) : std::runtime_error(what) {} };

// --bin-records 0: what a bin of the BAM stage is sized for when nothing is said (usage text, plan and run agree on this one number)
constexpr unsigned long long DEFAULT_BIN_RECORDS = 4000000ULL;

struct AlignOptions
{
    enum Action { RUN, HELP, VERSION };
    Action action = RUN;
    std::vector<std::string> argv;                          // @PG CL
    std::vector<std::string> baseCalls;                     // -b, one flowcell each
    std::vector<std::string> defaultAdapters;               // --default-adapters, one per flowcell (this host: all the same)
    std::vector<std::string> baseCallsFormat;               // fastq | fastq-gz per flowcell (the last one serves the rest)
    std::string referenceGenome;                            // -r sorted-reference.xml
    std::string outputDirectory = "./Aligned";              // -o
    std::string tempDirectory = "./Temp";                   // -t: where the bins' parts go that neither the device nor --memory-limit gigabytes of host memory hold
    unsigned memoryLimit = 0;                               // -m, gigabytes of host memory for the bins' parts (0: as much as the host has)
    std::string seeds = "auto", gapScoring = "bwa", dodgyAlignmentScore = "0", keepUnaligned = "back", realignGaps = "sample", useBasesMask = "default";
    std::string bamPuFormat = "%F:%L:%B", description, bamExcludeTags = "ZX,ZY", tls;
    std::string devices;                                    // --devices 0,1,...: one worker (context + thread) per entry; empty: --device alone
    std::vector<std::string> bamHeaderTags;
    unsigned seedLength = 32, firstPassSeeds = 1, jobs = 0, repeatThreshold = 10, laneNumberMax = 8, clustersAtATime = 0, mapqThreshold = 0, baseQualityCutoff = 25,
             semialignedGapLimit = 100, gappedMismatches = 5, realignedGapsPerFragment = 1;
    unsigned binRecords = 0;                                // --bin-records: records a bin of the BAM stage is sized for (0: DEFAULT_BIN_RECORDS = 4 million); this host's stand-in for the reference's bin size from --memory-limit
    int shadowScanRange = -1, bamGzipLevel = 1, device = 0;
    bool ignoreNeighbors = false, perTileTls = false, scatterRepeats = false, clipSemialigned = true, clipOverlapping = true, realignVigorously = false, realignDodgy = false,
         keepDuplicates = true, markDuplicates = true, pessimisticMapQ = false, variableReadLength = false, variableFastqReadLength = false, allowEmptyFlowcells = false;

    static std::string usage();
    // parses and checks; throws InvalidOption
    static AlignOptions parse(int argc, char **argv);
    // the isaac_params these options stand for, for reads of the given lengths (0: no second read)
    isaac_params params(unsigned readLength1, unsigned readLength2) const;
    bool keepUnalignedRecords() const { return "discard" != keepUnaligned; }
    unsigned forcedDodgyAlignmentScore() const;             // the MAPQ of alignments whose score is unknown
    std::vector<int> deviceList() const;                    // --devices, or --device alone
};

} // namespace isaac_host
