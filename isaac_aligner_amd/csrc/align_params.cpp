// The part of options::AlignOptions that fills isaac_params (host code, no GPU work): defaults (lib/options/AlignOptions.cpp:77-160), the
// --gap-scoring presets and syntax (:55-56,689-743), the seed descriptors of --seeds (lib/options/alignOptions/SeedDescriptorOption.cpp:38-244)
// and the first-pass rule (AlignOptions.cpp:1165-1171).
#include "../../include/isaac_gpu.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

namespace
{
thread_local std::string g_paramsError;
int paramsFail(const std::string &what) { g_paramsError = what; return ISAAC_GPU_EINVAL; }

std::vector<std::string> split(const std::string &s, char separator)
{
    std::vector<std::string> parts(1);
    for (const char c : s) { if (c == separator) parts.push_back(std::string()); else parts.back().push_back(c); }
    return parts;
}
bool parseInt(const std::string &s, long &v)
{
    if (s.empty()) return false;
    char *end = 0;
    v = std::strtol(s.c_str(), &end, 10);
    return !*end;
}

// adds one seed; false when the list is full
bool addSeed(isaac_params &p, uint32_t offset, uint32_t readIndex)
{
    if (p.n_seeds >= ISAAC_GPU_MAX_SEEDS) return false;
    isaac_seed &s = p.seeds[p.n_seeds++];
    s.offset = uint16_t(offset); s.length = uint16_t(p.seed_length); s.read_index = readIndex;
    return true;
}

// each returns the number of first-pass seeds the read has room for (SeedDescriptorOption.cpp: parseManual / parseAuto / parseAllSeedDescriptor)
int manualSeeds(isaac_params &p, const std::string &descriptor, uint32_t readIndex, unsigned &made)
{
    made = 0;
    for (const std::string &offsetString : split(descriptor, ':'))
    {
        long offset = 0;
        if (!parseInt(offsetString, offset) || offset < 0) return paramsFail("\n   *** Invalid seed offset '" + offsetString + "' found in '" + descriptor + "' ***\n");
        if (offset + p.seed_length > p.read_length[readIndex]) continue;           // "ignored as it stretches beyond the read"
        if (!addSeed(p, uint32_t(offset), readIndex)) return paramsFail("more than " + std::to_string(ISAAC_GPU_MAX_SEEDS) + " seeds");
        ++made;
    }
    return 0;
}
int autoSeeds(isaac_params &p, uint32_t readIndex, unsigned &made)
{
    const uint32_t length = p.read_length[readIndex], seedLength = p.seed_length;
    uint32_t generated = 0, offset = 0, endOffset = length;
    made = 1;
    bool room = true;
    if (length > seedLength)
    {   // a seed at either end: the best chance to miss the homopolymers
        room = addSeed(p, 0, readIndex) && addSeed(p, length - seedLength, readIndex);
        offset = seedLength; endOffset = length - seedLength; generated = 2; made = 2;
    }
    for (; room && offset + seedLength <= endOffset; offset += seedLength, ++generated) room = addSeed(p, offset, readIndex);
    // fewer than four so far: overlapping ones, half a seed off
    offset = seedLength / 2;
    if (endOffset > seedLength / 2)
    {
        endOffset -= seedLength / 2;
        for (; room && generated < 4 && offset + seedLength <= endOffset; offset += seedLength, ++generated) room = addSeed(p, offset, readIndex);
    }
    return room ? 0 : paramsFail("more than " + std::to_string(ISAAC_GPU_MAX_SEEDS) + " seeds");
}
int allSeeds(isaac_params &p, uint32_t readIndex, unsigned &made)
{
    made = 0;
    if (p.read_length[readIndex] < p.seed_length) return paramsFail("Read is too short for seed length " + std::to_string(p.seed_length));
    for (uint32_t i = 0; i < p.read_length[readIndex] - p.seed_length; ++i, ++made)
        if (!addSeed(p, i, readIndex)) return paramsFail("more than " + std::to_string(ISAAC_GPU_MAX_SEEDS) + " seeds");
    return 0;
}
} // namespace

extern "C" {

const char *isaac_gpu_params_last_error(void) { return g_paramsError.c_str(); }

int isaac_gpu_parse_gap_scoring(const char *gapScoring, isaac_params *p)
{
    if (!gapScoring || !p) return paramsFail("gap_scoring and params are required");
    std::string text = gapScoring;
    if ("bwa" == text) text = "0:-3:-11:-4:-20"; else if ("eland" == text) text = "2:-1:-15:-3:-25";
    const std::vector<std::string> parts = split(text, ':');
    if (5 != parts.size()) return paramsFail("\n   *** The 'gap-scoring' string must contain five components delimited by ':' ***\n");
    long v[5];
    for (int i = 0; i < 5; ++i) if (!parseInt(parts[i], v[i])) return paramsFail("bad lexical cast: source type value could not be interpreted as target");
    if (0 > v[0]) return paramsFail("\n   *** The 'gap-scoring' string must contain positive value or 0 for match score ***\n");
    if (0 < v[1]) return paramsFail("\n   *** The 'gap-scoring' string must contain negative value or 0 for mismatch score ***\n");
    if (0 < v[2]) return paramsFail("\n   *** The 'gap-scoring' string must contain negative value or 0 for gap open score ***\n");
    if (0 < v[3]) return paramsFail("\n   *** The 'gap-scoring' string must contain negative value or 0 for gap extend score ***\n");
    if (0 < v[4]) return paramsFail("\n   *** The 'gap-scoring' string must contain negative value or 0 for gap extend score cap ***\n");
    p->gap_match = int32_t(v[0]); p->gap_mismatch = int32_t(v[1]); p->gap_open = int32_t(v[2]); p->gap_extend = int32_t(v[3]); p->min_gap_extend = int32_t(v[4]);
    return 0;
}

int isaac_gpu_parse_seeds(const char *descriptor, uint32_t firstPassSeeds, isaac_params *p)
{
    if (!descriptor || !p) return paramsFail("descriptor and params are required");
    if (!p->n_reads || p->n_reads > 2) return paramsFail("n_reads must be 1 or 2");
    const std::string text = descriptor;
    if (text.empty()) return paramsFail("\n   *** The seed descriptor is empty. At least one seed is needed ***\n");
    if (p->semialigned_gap_limit && "auto" == text) firstPassSeeds = 2;                    // AlignOptions.cpp:1165-1171
    std::vector<std::string> perRead = split(text, ',');
    if (perRead.size() > p->n_reads)
        return paramsFail("\n   *** Too many lists-of-seeds in seed-descriptor '" + text + "': found " + std::to_string(perRead.size()) + ": " + std::to_string(p->n_reads) + " reads only ***\n");
    perRead.resize(p->n_reads, perRead.back());                                              // the last list serves the reads that follow
    p->n_seeds = 0;
    for (uint32_t r = 0; r < p->n_reads; ++r)
    {
        const std::string &d = perRead[r];
        if (d.empty()) return paramsFail("\n   *** The seed descriptor for read " + std::to_string(r + 1) + " is empty. At least one seed is needed ***\n");
        unsigned made = 0;
        const int rc = "all" == d ? allSeeds(*p, r, made) : "auto" == d ? autoSeeds(*p, r, made) : manualSeeds(*p, d, r, made);
        if (rc) return rc;
        firstPassSeeds = std::min<uint32_t>(firstPassSeeds, made);
    }
    if (0 >= firstPassSeeds) return paramsFail("\n   *** At least one seed must be used on the first pass (--first-pass-seeds is " + std::to_string(firstPassSeeds) + ") ***\n");
    p->first_pass_seeds = firstPassSeeds;
    return 0;
}

// flowcell::SequencingAdapterListGrammar (include/flowcell/SequencingAdapterListGrammar.hpp:52-104) by hand: start_ = macro_ | adapter_list_;
// adapter_list_ = *(adapter_ >> -','); adapter_ = sequence '*' | sequence | '*' sequence, a sequence being five or more of ACGTacgt (stored upper case).
// What is left unparsed is an error (DefaultAdaptersOption.cpp:42-47).
int isaac_gpu_parse_adapters(const char *descriptor, isaac_params *p)
{
    if (!descriptor || !p) return paramsFail("descriptor and params are required");
    const std::string text = descriptor;
    struct Preset { const char *name; const char *first; bool firstReverse; bool bounded; const char *second; bool secondReverse; };
    // lib/flowcell/SequencingAdapterMetadata.cpp:29-39; the grammar tries Standard, NexteraMp, Nextera in this order
    static const Preset presets[] = { { "Standard", "AGATCGGAAGAGC", false, false, "GCTCTTCCGATCT", true },
                                      { "NexteraMp", "CTGTCTCTTATACACATCT", false, true, "AGATGTGTATAAGAGACAG", false },
                                      { "Nextera", "CTGTCTCTTATACACATCT", false, false, "AGATGTGTATAAGAGACAG", true } };
    std::vector<isaac_adapter> list;
    size_t at = 0;
    const auto add = [&](const std::string &sequence, bool reverse, bool bounded)
    {
        isaac_adapter a; std::memset(&a, 0, sizeof(a));
        std::memcpy(a.sequence, sequence.data(), std::min(sequence.size(), sizeof(a.sequence) - 1));
        a.reverse = reverse ? 1 : 0; a.clip_length = bounded ? uint32_t(sequence.size()) : 0;
        list.push_back(a);
        return sequence.size() < sizeof(a.sequence) - 1;
    };
    bool macro = false, tooLong = false;
    for (const Preset &preset : presets)
        if (0 == text.compare(0, std::strlen(preset.name), preset.name))
        {
            add(preset.first, preset.firstReverse, preset.bounded); add(preset.second, preset.secondReverse, preset.bounded);
            at = std::strlen(preset.name); macro = true;
            break;
        }
    const auto sequenceAt = [&](size_t from, std::string &sequence)
    {
        sequence.clear();
        size_t i = from;
        for (; i < text.size(); ++i)
        {
            const char c = text[i];
            if ('A' == c || 'a' == c) sequence.push_back('A'); else if ('C' == c || 'c' == c) sequence.push_back('C');
            else if ('G' == c || 'g' == c) sequence.push_back('G'); else if ('T' == c || 't' == c) sequence.push_back('T'); else break;
        }
        return sequence.size() >= 5 ? i : from;
    };
    while (!macro && at < text.size())
    {
        std::string sequence;
        size_t end = sequenceAt(at, sequence);
        if (end != at)
        {
            if (end < text.size() && '*' == text[end]) { tooLong |= !add(sequence, false, false); ++end; }    // forward_unbounded_adapter_
            else tooLong |= !add(sequence, false, true);                                                       // simple_adapter_
        }
        else if ('*' == text[at] && (end = sequenceAt(at + 1, sequence)) != at + 1) tooLong |= !add(sequence, true, false);   // reverse_unbounded_adapter_
        else break;
        at = end;
        if (at < text.size() && ',' == text[at]) ++at;
    }
    if (at != text.size()) return paramsFail("\n   *** Could not parse the default-adapters '" + text + "' at: " + text.substr(at) + " ***\n");
    if (tooLong) return paramsFail("Adapter sequence is too long");                                             // SequencingAdapter.cpp:35
    if (list.size() > ISAAC_GPU_MAX_ADAPTERS) return paramsFail("more than " + std::to_string(ISAAC_GPU_MAX_ADAPTERS) + " sequencing adapters");
    p->n_adapters = uint32_t(list.size());
    std::memset(p->adapters, 0, sizeof(p->adapters));
    for (size_t i = 0; i < list.size(); ++i) p->adapters[i] = list[i];
    return 0;
}

int isaac_gpu_default_params(uint32_t readLength1, uint32_t readLength2, isaac_params *p)
{
    if (!p || !readLength1) return paramsFail("params and read_length1 are required");
    std::memset(p, 0, sizeof(*p));
    p->repeat_threshold = 10; p->gapped_mismatches_max = 5; p->semialigned_gap_limit = 100; p->base_quality_cutoff = 25;
    p->ignore_neighbors = 0; p->clip_semialigned = 1; p->clip_overlapping = 1; p->scatter_repeats = 0; p->dodgy_alignment_score = 0;
    p->mapq_threshold = 0; p->keep_unaligned = 1; p->mate_drift_range = -1; p->seed_length = 32;
    p->n_reads = readLength2 ? 2 : 1; p->read_length[0] = readLength1; p->read_length[1] = readLength2;
    const int rc = isaac_gpu_parse_gap_scoring("bwa", p);
    return rc ? rc : isaac_gpu_parse_seeds("auto", 1, p);
}

} // extern "C"
