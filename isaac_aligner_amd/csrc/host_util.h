// Host-side preparation shared by the product library (capi.hip) and the CPU debugging harness (tests/hostemu):
// parameter expansion, the quality lookup tables, the rest-of-genome correction and the sequential template-length
// learning.  These are the pieces of the path whose arithmetic depends on glibc (pow/log/exp), so they are evaluated on
// the host once and handed to the kernels as plain numbers.
#pragma once
#include "../../include/isaac_gpu.h"
#include "template.h"
#include <cmath>
#include <cstring>
#include <limits>
#include <algorithm>
#include <string>
#include <vector>
#include <stdexcept>

namespace isaac
{

// AlignerBase.cpp:32-44, FindMatchesTransition.cpp:90-110 (seed index lists per iteration, each ordered by (read, offset) as
// SeedGeneratorBase.cpp:34-38 does), flowcell::ReadMetadata geometry
inline DevParams makeDevParams(const isaac_params &p)
{
    if (p.n_reads < 1 || p.n_reads > 2) throw std::invalid_argument("n_reads must be 1 or 2");
    if (p.seed_length != 32) throw std::invalid_argument("only --seed-length 32 is implemented");
    if (!p.n_seeds || p.n_seeds > MAX_SEEDS) throw std::invalid_argument("1..16 seeds are supported");
    if (!p.repeat_threshold || p.repeat_threshold > 16) throw std::invalid_argument("repeat_threshold must be in 1..16");
    DevParams d;
    std::memset(&d, 0, sizeof(d));
    d.gapMatch = p.gap_match; d.gapMismatch = p.gap_mismatch; d.gapOpen = p.gap_open; d.gapExtend = p.gap_extend; d.minGapExtend = p.min_gap_extend;
    d.normalizedMismatchScore = u32(p.gap_match - p.gap_mismatch); d.normalizedGapOpenScore = u32(p.gap_match - p.gap_open);
    d.normalizedGapExtendScore = u32(p.gap_match - p.gap_extend); d.normalizedMaxGapExtendScore = u32(-p.min_gap_extend);
    d.repeatThreshold = p.repeat_threshold; d.gappedMismatchesMax = p.gapped_mismatches_max; d.semialignedGapLimit = p.semialigned_gap_limit;
    d.baseQualityCutoff = p.base_quality_cutoff; d.ignoreNeighbors = p.ignore_neighbors; d.clipSemialigned = p.clip_semialigned; d.clipOverlapping = p.clip_overlapping;
    d.scatterRepeats = p.scatter_repeats; d.dodgyAlignmentScore = p.dodgy_alignment_score; d.mapqThreshold = p.mapq_threshold; d.keepUnaligned = p.keep_unaligned;
    d.mateDriftRange = p.mate_drift_range;
    d.nReads = p.n_reads;
    u32 offset = 0, firstCycle = 1;
    for (u32 r = 0; r < p.n_reads; ++r)
    {
        if (!p.read_length[r] || p.read_length[r] > 512) throw std::invalid_argument("read lengths 1..512 are supported");
        d.readLength[r] = p.read_length[r]; d.readOffset[r] = offset; d.firstCycle[r] = firstCycle;
        offset += p.read_length[r]; firstCycle += p.read_length[r];
    }
    d.clusterLength = offset;
    d.nSeeds = p.n_seeds;
    u32 countsPerRead[2] = { 0, 0 };
    for (u32 s = 0; s < p.n_seeds; ++s)
    {
        const isaac_seed &seed = p.seeds[s];
        if (seed.read_index >= p.n_reads || seed.length != 32 || u32(seed.offset) + seed.length > p.read_length[seed.read_index]) throw std::invalid_argument("bad seed");
        d.seeds[s].offset = seed.offset; d.seeds[s].length = seed.length; d.seeds[s].readIndex = seed.read_index;
        const u32 iteration = (p.first_pass_seeds > countsPerRead[seed.read_index]) ? 0 : 1;
        d.passSeeds[iteration][d.nPass[iteration]++] = u8(s);
        ++countsPerRead[seed.read_index];
    }
    d.maxSeedsPerRead = std::max(countsPerRead[0], countsPerRead[1]);
    for (u32 it = 0; it < 2; ++it)
        std::stable_sort(d.passSeeds[it], d.passSeeds[it] + d.nPass[it], [&](u8 a, u8 b)
        { return d.seeds[a].readIndex < d.seeds[b].readIndex || (d.seeds[a].readIndex == d.seeds[b].readIndex && d.seeds[a].offset < d.seeds[b].offset); });
    if (2 * d.nSeeds * (d.repeatThreshold > 1 ? d.repeatThreshold - 1 : 1) > MATCH_CAP_MAX) throw std::invalid_argument("seeds x repeat threshold exceeds the match capacity");
    return d;
}

// matchSelector::SequencingAdapter's constructor (lib/alignment/matchSelector/SequencingAdapter.cpp:30-56) for every adapter of the list: the position of each
// 5-mer of the adapter, -1 where there is none, -2 where there are several.  n == 0: no adapters (DevParams::adapters stays NULL).
inline DevAdapters makeDevAdapters(const isaac_params &p)
{
    DevAdapters d;
    std::memset(&d, 0, sizeof(d));
    if (p.n_adapters > MAX_ADAPTERS) throw std::invalid_argument("at most 8 sequencing adapters are supported");
    d.n = p.n_adapters;
    for (u32 k = 0; k < p.n_adapters; ++k)
    {
        const isaac_adapter &in = p.adapters[k];
        DevAdapter &a = d.a[k];
        const size_t length = strnlen(in.sequence, sizeof(in.sequence));
        if (length < ADAPTER_MATCH_BASES_MIN) throw std::invalid_argument("a sequencing adapter has at least 5 bases");              // SequencingAdapterListGrammar.hpp: five adapter_char_ at least
        if (length > MAX_ADAPTER_LENGTH) throw std::invalid_argument("Adapter sequence is too long");                                 // SequencingAdapter.cpp:35
        if (in.clip_length && in.clip_length < length) throw std::invalid_argument("Clip length cannot be shorter than the adapter sequence");   // :36-38
        for (size_t i = 0; i < length; ++i) if (!std::strchr("ACGT", in.sequence[i])) throw std::invalid_argument("adapter sequences are made of A, C, G and T");
        a.length = u32(length); a.reverse = in.reverse ? 1 : 0; a.clipLength = in.clip_length;
        std::memcpy(a.sequence, in.sequence, length);
        std::memset(a.kmerPositions, -1, sizeof(a.kmerPositions));
        u32 kmer = 0;
        for (size_t i = 0; i < length; ++i)
        {
            kmer = ((kmer << 2) | u32(std::strchr("ACGT", in.sequence[i]) - "ACGT")) & 1023u;
            if (i + 1 < ADAPTER_MATCH_BASES_MIN) continue;
            signed char &pos = a.kmerPositions[kmer];
            if (-1 == pos) pos = (signed char)(i + 1 - ADAPTER_MATCH_BASES_MIN);
            else if (-2 != pos) pos = -2;
        }
    }
    return d;
}

// lib/alignment/Quality.cpp:34-66: 100 entries each; entry 0 of BOTH tables is log(1 - 10^-0.1)
inline void makeQualityTables(double *logMatch, double *logMismatch)
{
    logMatch[0] = log(1.0 - pow(10.0, 1.0 / -10.0));
    for (int i = 1; i < 100; ++i) logMatch[i] = log(1.0 - pow(10.0, (double)i / -10.0));
    logMismatch[0] = log(1.0 - pow(10.0, 1.0 / -10.0));
    for (unsigned q = 1; q < 100U; ++q) logMismatch[q] = log(pow(10.0, (double)q / -10.0) / 3.0);
}
inline double logMismatchQ40() { return log(pow(10.0, (double)40 / -10.0) / 3.0); }

// RestOfGenomeCorrection.hh:44-88 + Quality.hh:87-91 + reference/Contig.cpp:30-38 over the loaded contigs
inline RogCorrection makeRogCorrection(const DevParams &P, const u64 *contigOffsets, const u8 *contigLoaded, u32 nContigs)
{
    size_t genomeLength = 0;
    for (u32 c = 0; c < nContigs; ++c) if (!contigLoaded || contigLoaded[c]) genomeLength += size_t(contigOffsets[c + 1] - contigOffsets[c]);
    RogCorrection r; r.read[0] = r.read[1] = 0.0;
    unsigned total = 0;
    for (u32 i = 0; i < P.nReads; ++i)
    {
        r.read[i] = std::max(exp(log(2.0) + log((double)unsigned(genomeLength)) - (log(4.0) * (double)P.readLength[i])), std::numeric_limits<double>::min());
        total += P.readLength[i];
    }
    r.pair = std::max(exp(log(2.0) + log((double)unsigned(genomeLength)) - (log(4.0) * (double)total)), std::numeric_limits<double>::min());
    return r;
}

// TemplateLengthDistribution (lib/alignment/TemplateLengthStatistics.cpp:105-159,275-358): the statistics are learnt
// sequentially, in cluster order, from clusters that have exactly one candidate per read; the per-cluster facts come from the
// fragment kernel run without gaps and without trimming (MatchSelector.cpp:188-256).

struct TlsLearner
{
    DevTls stats; int mateDriftRange;
    std::vector<unsigned> lengthList; std::vector<std::vector<unsigned> > histograms; unsigned count;
    explicit TlsLearner(int drift) : mateDriftRange(drift), histograms(8), count(0) { clear(); }
    void clear()
    {
        stats.min = stats.max = stats.median = stats.lowStdDev = stats.highStdDev = 0xffffffffu; stats.bestModel[0] = stats.bestModel[1] = 8;
        stats.stable = 0; stats.mateMin = stats.mateMax = 0xffffffffu; count = 0;
        for (auto &h : histograms) h.clear();
        lengthList.clear();
    }
    void setMin(unsigned v) { stats.min = v; stats.mateMin = -1 == mateDriftRange ? stats.min : stats.median - mateDriftRange; }
    void setMedian(unsigned v) { stats.median = v; stats.mateMin = -1 == mateDriftRange ? stats.min : stats.median - mateDriftRange; stats.mateMax = -1 == mateDriftRange ? stats.max : stats.median + mateDriftRange; }
    void setMax(unsigned v) { stats.max = v; stats.mateMax = -1 == mateDriftRange ? stats.max : stats.median + mateDriftRange; }
    static bool sameFive(const DevTls &a, const DevTls &b)
    { return a.min == b.min && a.median == b.median && a.max == b.max && a.lowStdDev == b.lowStdDev && a.highStdDev == b.highStdDev; }
    void updateStatistics()
    {
        static const double CI = std::erf(3.0 / std::sqrt(2.0)), CI1 = std::erf(1.0 / std::sqrt(2.0));
        static const double LOWER = (1.0 - CI) / 2.0, UPPER = (1.0 + CI) / 2.0, LOWER1 = (1.0 - CI1) / 2.0, UPPER1 = (1.0 + CI1) / 2.0;
        const DevTls old = stats;
        stats.bestModel[0] = histograms[1].size() <= histograms[0].size() ? 0 : 1;
        stats.bestModel[1] = (stats.bestModel[0] + 1) % 2;
        for (size_t i = 2; histograms.size() > i; ++i)
        {
            if (histograms[i].size() > histograms[stats.bestModel[0]].size()) { stats.bestModel[1] = stats.bestModel[0]; stats.bestModel[0] = int(i); }
            else if (histograms[i].size() > histograms[stats.bestModel[1]].size()) stats.bestModel[1] = int(i);
        }
        lengthList.clear();
        lengthList.insert(lengthList.end(), histograms[stats.bestModel[0]].begin(), histograms[stats.bestModel[0]].end());
        lengthList.insert(lengthList.end(), histograms[stats.bestModel[1]].begin(), histograms[stats.bestModel[1]].end());
        std::sort(lengthList.begin(), lengthList.end());
        setMin(lengthList.empty() ? 0 : lengthList[unsigned(lengthList.size() * LOWER)]);
        setMedian(lengthList.empty() ? TEMPLATE_LENGTH_THRESHOLD / 2 : lengthList[unsigned(lengthList.size() * 0.5)]);
        setMax(lengthList.empty() ? TEMPLATE_LENGTH_THRESHOLD : lengthList[unsigned(lengthList.size() * UPPER)]);
        stats.lowStdDev = lengthList.empty() ? stats.median : (stats.median - lengthList[unsigned(lengthList.size() * LOWER1)]);
        stats.highStdDev = lengthList.empty() ? stats.median : (lengthList[unsigned(lengthList.size() * UPPER1)] - stats.median);
        if (sameFive(old, stats) && old.bestModel[0] == stats.bestModel[0] && old.bestModel[1] == stats.bestModel[1]) stats.stable = 1;
    }
    // addTemplate (TemplateLengthStatistics.cpp:275-340)
    bool add(const TlsSample &s)
    {
        if (!s.valid || !s.n0 || !s.n1) return stats.stable;
        if (1 < s.n0 || 1 < s.n1) return stats.stable;
        if (s.contig0 != s.contig1) return stats.stable;
        if (s.insertEnd) return stats.stable;
        Cand a, b; candInit(a, 0); candInit(b, 1);
        a.contigId = s.contig0; a.position = s.pos0; a.observedLength = s.obs0; a.reverse = s.rev0; a.cigarLength = 1;
        b.contigId = s.contig1; b.position = s.pos1; b.observedLength = s.obs1; b.reverse = s.rev1; b.cigarLength = 1;
        const u64 length = tlsGetLength(a, b);
        if (length > TEMPLATE_LENGTH_THRESHOLD) return stats.stable;
        const i32 am = tlsAlignmentModel(a, b);
        if (8 != am)
        {
            histograms[am].push_back(unsigned(length));
            ++count;
            if (0 == (count % 10000))
            {
                const DevTls old = stats;
                updateStatistics();
                if (sameFive(old, stats)) stats.stable = 1;
            }
        }
        return stats.stable;
    }
    bool finalize()
    {
        const DevTls old = stats;
        updateStatistics();
        if (sameFive(old, stats)) stats.stable = 1;
        return stats.stable;
    }
};

} // namespace isaac
