// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Seed descriptor, seed generation, the reference's sort-merge-join match finder and the match sort.
// Parity unpinned by reference tests (nothing in the reference tests MatchFinder/ExactMaskMatcher directly).
#include "oracle.hpp"

#include <omp.h>
#include <parallel/algorithm>
#include <algorithm>
#include <stdexcept>
#include <map>
#include <cstring>
#include <thread>
#include <string>

namespace oracle
{

// How a thread moves through the table between two seed k-mers: 0 = entry by entry (the merge join of ExactMaskMatcher.cpp:83-184, which
// streams the whole table per pass and is what the reference does with its batches of millions of clusters), 1 = by bisection of what is left
// (the same matches; less traffic when the seeds are few against the table).  Set by oracle_set_lookup_mode; bench.py times both.
static int g_lookupMode = 0;


// lib/options/alignOptions/SeedDescriptorOption.cpp:90-151
unsigned parseAutoSeedDescriptor(bool /*detectSimpleIndels*/, const ReadMetadata &read, unsigned seedLength, std::vector<SeedMetadata> &out)
{
    unsigned ret = 1, generated = 0, offset = 0, endOffset = read.length;
    if (read.length > seedLength)
    {
        SeedMetadata s0 = { 0, seedLength, read.index, unsigned(out.size()) }; out.push_back(s0);
        offset = seedLength;
        endOffset = read.length - seedLength;
        SeedMetadata s1 = { endOffset, seedLength, read.index, unsigned(out.size()) }; out.push_back(s1);
        generated = 2; ret = 2;
    }
    while (offset + seedLength <= endOffset)
    {
        SeedMetadata s = { offset, seedLength, read.index, unsigned(out.size()) }; out.push_back(s);
        ++generated; offset += seedLength;
    }
    offset = seedLength / 2;
    if (endOffset > seedLength / 2)
    {
        endOffset -= seedLength / 2;
        while (generated < 4 && offset + seedLength <= endOffset)
        {
            SeedMetadata s = { offset, seedLength, read.index, unsigned(out.size()) }; out.push_back(s);
            ++generated; offset += seedLength;
        }
    }
    return ret;
}

// SeedDescriptorOption.cpp:209-245
std::vector<SeedMetadata> autoSeeds(bool detectSimpleIndels, const std::vector<ReadMetadata> &reads, unsigned seedLength, unsigned &firstPassSeeds)
{
    std::vector<SeedMetadata> ret;
    for (size_t i = 0; i < reads.size(); ++i)
        firstPassSeeds = std::min(firstPassSeeds, parseAutoSeedDescriptor(detectSimpleIndels, reads[i], seedLength, ret));
    return ret;
}

// lib/workflow/alignWorkflow/FindMatchesTransition.cpp:90-110
std::vector<std::vector<unsigned> > seedIndexListPerIteration(const std::vector<SeedMetadata> &seeds, unsigned nReads, unsigned firstPassSeeds)
{
    std::vector<std::vector<unsigned> > ret(2);
    std::vector<unsigned> countsPerRead(nReads, 0);
    for (size_t i = 0; i < seeds.size(); ++i)
    {
        const unsigned iteration = (firstPassSeeds > countsPerRead[seeds[i].readIndex]) ? 0 : 1;
        ret[iteration].push_back(seeds[i].index);
        ++countsPerRead[seeds[i].readIndex];
    }
    while (!ret.empty() && ret.back().empty()) ret.resize(ret.size() - 1);
    return ret;
}

// AlignOptions.cpp:77-160 defaults, :1165-1171 (firstPassSeeds = 2 for "auto" when semialigned-gap-limit != 0)
Params makeParams(unsigned nReads, unsigned len1, unsigned len2)
{
    Params p;
    unsigned firstCycle = 1, offset = 0;
    const unsigned lens[2] = { len1, len2 };
    for (unsigned r = 0; r < nReads; ++r)
    {
        ReadMetadata rm = { lens[r], r, offset, firstCycle };
        p.reads.push_back(rm);
        offset += lens[r]; firstCycle += lens[r];
    }
    p.firstPassSeeds = p.semialignedGapLimit ? 2 : 1;
    p.seeds = autoSeeds(0 != p.semialignedGapLimit, p.reads, p.seedLength, p.firstPassSeeds);
    return p;
}

// include/alignment/SeedMetadata.hh:103-108
static bool firstCycleLess(const SeedMetadata &l, const SeedMetadata &r)
{ return l.readIndex < r.readIndex || (l.readIndex == r.readIndex && l.offset < r.offset); }

// include/alignment/Seed.hh:88-93
static bool orderByKmerSeedIndex(const Seed &l, const Seed &r)
{ return l.kmer < r.kmer || (l.kmer == r.kmer && SeedId(l.seedId).getSeed() < SeedId(r.seedId).getSeed()); }

// lib/alignment/ClusterSeedGenerator.cpp:138-192 (one barcode, index 0) + SeedGeneratorBase.cpp:71-94
void generateSeeds(const Params &p, const std::vector<unsigned> &seedIndexList, const uint8_t *bcl, unsigned nClusters,
                   unsigned tile, const ClusterInfo &complete, std::vector<Seed> &seeds)
{
    std::vector<SeedMetadata> ordered;
    for (size_t i = 0; i < seedIndexList.size(); ++i) ordered.push_back(p.seeds.at(seedIndexList[i]));
    std::sort(ordered.begin(), ordered.end(), firstCycleLess);
    const unsigned clusterLength = p.clusterLength();
    seeds.clear();
    for (unsigned clusterId = 0; clusterId < nClusters; ++clusterId)
    {
        const uint8_t *clusterIt = bcl + size_t(clusterId) * clusterLength;
        for (size_t s = 0; s < ordered.size(); ++s)
        {
            const SeedMetadata &sm = ordered[s];
            if (complete[clusterId].isReadComplete(sm.readIndex)) continue;
            Seed fw = { 0, SeedId(tile, 0, clusterId, sm.index, 0).value };
            Seed rv = { 0, SeedId(tile, 0, clusterId, sm.index, 1).value };
            const uint8_t *baseIt = clusterIt + sm.offset + p.reads.at(sm.readIndex).offset;
            for (unsigned len = 32; len; --len, ++baseIt)
            {
                const uint8_t base = *baseIt;
                if (base & 0xfc)
                {
                    const uint64_t f = base & 3, r = (~f) & 3;
                    fw.kmer <<= 2; fw.kmer |= f;
                    rv.kmer >>= 2; rv.kmer |= r << 62;
                }
                else
                {
                    // Seed.hh:80-85 makeNSeed(tile, barcode, cluster, lowestSeedId): reverse bit = !lowest
                    const bool lowest = (0 == sm.index);
                    fw.kmer = ~uint64_t(0); fw.seedId = SeedId(tile, 0, clusterId, SeedId::SEED_MASK, !lowest).value;
                    rv = fw;
                    break;
                }
            }
            seeds.push_back(fw); seeds.push_back(rv);
        }
    }
    std::sort(seeds.begin(), seeds.end(), orderByKmerSeedIndex);
}

// lib/alignment/matchFinder/ExactMaskMatcher.cpp:83-184 over the concatenation of all mask files, plus the N-seed handling of
// MatchFinder.cpp:213-249.  The reference reads `isReadComplete` in generateNoMatches while other mask threads may be
// setting it (benign race); this restatement resolves it by evaluating NoMatch emission after the whole pass.
void findMatchesExact(const Params &p, const SortedReference &ref, const std::vector<Seed> &seeds, bool closeRepeats, bool storeNoMatches,
                      ClusterInfo &complete, std::vector<Match> &out, std::vector<uint8_t> &contigHasMatches)
{
    const ReferencePosition tooMany(ReferencePosition::TooManyMatch), noMatch(ReferencePosition::NoMatch);
    std::vector<Seed> pendingNoMatch;
    size_t nextSeed = 0, nextRef = 0;
    const size_t nRef = ref.kmers.size();
    std::vector<ReferenceKmer> repeatList;
    // seeds are sorted by (kmer, seed index); N-seeds (kmer ~0, seed index 255) are at the very end
    size_t endSeeds = seeds.size();
    while (endSeeds && SeedId(seeds[endSeeds - 1].seedId).isNSeedId()) --endSeeds;
    while (endSeeds != nextSeed)
    {
        const size_t currentSeed = nextSeed;
        while (endSeeds != nextSeed && seeds[currentSeed].kmer == seeds[nextSeed].kmer) ++nextSeed;
        if (g_lookupMode && nextRef < nRef && seeds[currentSeed].kmer > ref.kmers[nextRef].kmer)
            nextRef = size_t(std::lower_bound(ref.kmers.begin() + nextRef, ref.kmers.end(), seeds[currentSeed].kmer, [](const ReferenceKmer &r, uint64_t k) { return r.kmer < k; }) - ref.kmers.begin());
        while (nextRef < nRef && seeds[currentSeed].kmer > ref.kmers[nextRef].kmer) ++nextRef;
        repeatList.clear();
        while (nextRef < nRef && seeds[currentSeed].kmer == ref.kmers[nextRef].kmer)
        {
            if (repeatList.size() < p.repeatThreshold)
            {
                ReferenceKmer rk = { ref.kmers[nextRef].kmer, ReferencePosition::fromValue(ref.kmers[nextRef].position).translateContig(ref.karyotype).value };
                repeatList.push_back(rk);
            }
            ++nextRef;
        }
        if (repeatList.empty())
        {
            if (storeNoMatches) for (size_t s = currentSeed; s < nextSeed; ++s) pendingNoMatch.push_back(seeds[s]);
        }
        else if (repeatList.size() >= p.repeatThreshold || ReferencePosition::fromValue(repeatList.front().position).isTooManyMatch())
        {
            for (size_t s = currentSeed; s < nextSeed; ++s)
            {
                const SeedId id(seeds[s].seedId);
                Match m = { id.value, tooMany.value }; out.push_back(m);
                if (closeRepeats) complete[id.getCluster()].markReadComplete(p.seeds[id.getSeed()].readIndex);
            }
        }
        else
        {
            const ReferencePosition anyPosition = ReferencePosition::fromValue(repeatList.front().position);
            for (size_t s = currentSeed; s < nextSeed; ++s)
            {
                const SeedId id(seeds[s].seedId);
                for (size_t r = 0; r < repeatList.size(); ++r) { Match m = { id.value, repeatList[r].position }; out.push_back(m); }
                if (p.ignoreNeighbors || !anyPosition.hasNeighbors()) complete[id.getCluster()].markReadComplete(p.seeds[id.getSeed()].readIndex);
            }
            for (size_t r = 0; r < repeatList.size(); ++r) // MatchDistribution::addMatches (>= 1 per repeat) -> "contig has matches"
                contigHasMatches.at(ReferencePosition::fromValue(repeatList[r].position).getContigId()) = 1;
        }
    }
    for (size_t i = 0; i < pendingNoMatch.size(); ++i)
    {
        const SeedId id(pendingNoMatch[i].seedId);
        if (!complete[id.getCluster()].isReadComplete(p.seeds[id.getSeed()].readIndex)) { Match m = { id.value, noMatch.value }; out.push_back(m); }
    }
    if (storeNoMatches) // MatchFinder.cpp:231-246
        for (size_t s = endSeeds; s < seeds.size(); ++s) { Match m = { seeds[s].seedId, noMatch.value }; out.push_back(m); }
}

// lib/workflow/alignWorkflow/SelectMatchesTransition.cpp:242-254
bool sortByTileBarcodeClusterLocation(const Match &l, const Match &r)
{
    const SeedId ls(l.seedId), rs(r.seedId);
    return ls.getTileBarcodeCluster() < rs.getTileBarcodeCluster() ||
        (ls.getTileBarcodeCluster() == rs.getTileBarcodeCluster() && (l.location < r.location ||
            (l.location == r.location && ls.getSeed() < rs.getSeed())));
}

// FindMatchesTransition.cpp:391-427 for one tile, default options (no neighbor pass).  The final sort adds the reverse bit as
// the last key: the reference's parallelSort leaves the order of (cluster, location, seed)-equal matches unspecified.
void findTileMatches(const Params &p, const SortedReference &ref, const uint8_t *bcl, unsigned nClusters, unsigned tile,
                     std::vector<Match> &matches, std::vector<uint8_t> &contigHasMatches)
{
    const std::vector<std::vector<unsigned> > perIteration = seedIndexListPerIteration(p.seeds, unsigned(p.reads.size()), p.firstPassSeeds);
    ClusterInfo complete(nClusters);      // nothing complete, no barcode (TileClusterInfo.hh:177-184)
    matches.clear();
    std::vector<Seed> seeds;
    generateSeeds(p, perIteration.at(0), bcl, nClusters, tile, complete, seeds);
    findMatchesExact(p, ref, seeds, false, 1 == perIteration.size(), complete, matches, contigHasMatches);
    if (2 == perIteration.size())
    {
        generateSeeds(p, perIteration.at(1), bcl, nClusters, tile, complete, seeds);
        findMatchesExact(p, ref, seeds, true, true, complete, matches, contigHasMatches);
    }
    std::sort(matches.begin(), matches.end(), [](const Match &l, const Match &r)
    {
        if (sortByTileBarcodeClusterLocation(l, r)) return true;
        if (sortByTileBarcodeClusterLocation(r, l)) return false;
        return (l.seedId & 1) < (r.seedId & 1);
    });
}

// ---------------------------------------------------------------- the same, on several threads
// MatchFinder::matchMaskParallel (lib/alignment/MatchFinder.cpp:251-316): every thread takes whole masks, i.e. a range of the
// k-mer space -- its share of the sorted seeds and the part of the sorted reference they can meet -- so the table is streamed
// once per iteration in total, not once per thread.  Read-complete marks are collected per thread and applied when all threads
// have finished: the merge join only reads them after the pass (NoMatch emission), which is also how the serial form resolves
// the reference's benign race.
namespace
{
struct PartialFind
{
    std::vector<Match> matches; std::vector<Seed> pendingNoMatch; std::vector<std::pair<unsigned, unsigned> > marks; std::vector<uint8_t> hits;
};

void joinRange(const Params &p, const SortedReference &ref, const std::vector<Seed> &seeds, size_t nextSeed, const size_t endSeeds, bool closeRepeats, bool storeNoMatches, PartialFind &out)
{
    const ReferencePosition tooMany(ReferencePosition::TooManyMatch);
    const size_t nRef = ref.kmers.size();
    if (nextSeed == endSeeds) return;
    // where a mask file that starts with this thread's first k-mer would begin
    size_t nextRef = size_t(std::lower_bound(ref.kmers.begin(), ref.kmers.end(), seeds[nextSeed].kmer,
                                             [](const ReferenceKmer &r, uint64_t k) { return r.kmer < k; }) - ref.kmers.begin());
    std::vector<ReferenceKmer> repeatList;
    while (endSeeds != nextSeed)
    {
        const size_t currentSeed = nextSeed;
        while (endSeeds != nextSeed && seeds[currentSeed].kmer == seeds[nextSeed].kmer) ++nextSeed;
        if (g_lookupMode && nextRef < nRef && seeds[currentSeed].kmer > ref.kmers[nextRef].kmer)
            nextRef = size_t(std::lower_bound(ref.kmers.begin() + nextRef, ref.kmers.end(), seeds[currentSeed].kmer, [](const ReferenceKmer &r, uint64_t k) { return r.kmer < k; }) - ref.kmers.begin());
        while (nextRef < nRef && seeds[currentSeed].kmer > ref.kmers[nextRef].kmer) ++nextRef;
        repeatList.clear();
        while (nextRef < nRef && seeds[currentSeed].kmer == ref.kmers[nextRef].kmer)
        {
            if (repeatList.size() < p.repeatThreshold)
            {
                ReferenceKmer rk = { ref.kmers[nextRef].kmer, ReferencePosition::fromValue(ref.kmers[nextRef].position).translateContig(ref.karyotype).value };
                repeatList.push_back(rk);
            }
            ++nextRef;
        }
        if (repeatList.empty())
        {
            if (storeNoMatches) for (size_t s = currentSeed; s < nextSeed; ++s) out.pendingNoMatch.push_back(seeds[s]);
        }
        else if (repeatList.size() >= p.repeatThreshold || ReferencePosition::fromValue(repeatList.front().position).isTooManyMatch())
        {
            for (size_t s = currentSeed; s < nextSeed; ++s)
            {
                const SeedId id(seeds[s].seedId);
                Match m = { id.value, tooMany.value }; out.matches.push_back(m);
                if (closeRepeats) out.marks.push_back(std::make_pair(unsigned(id.getCluster()), p.seeds[id.getSeed()].readIndex));
            }
        }
        else
        {
            const ReferencePosition anyPosition = ReferencePosition::fromValue(repeatList.front().position);
            for (size_t s = currentSeed; s < nextSeed; ++s)
            {
                const SeedId id(seeds[s].seedId);
                for (size_t r = 0; r < repeatList.size(); ++r) { Match m = { id.value, repeatList[r].position }; out.matches.push_back(m); }
                if (p.ignoreNeighbors || !anyPosition.hasNeighbors()) out.marks.push_back(std::make_pair(unsigned(id.getCluster()), p.seeds[id.getSeed()].readIndex));
            }
            for (size_t r = 0; r < repeatList.size(); ++r) out.hits.at(ReferencePosition::fromValue(repeatList[r].position).getContigId()) = 1;
        }
    }
}

// seeds of one iteration in (kmer, seed index) order: the clusters are cut into ranges, every thread extracts and sorts its own,
// the sorted pieces are merged pairwise (SeedGeneratorBase.cpp:182-206 sorts with parallelSort)
void generateSeedsParallel(const Params &p, const std::vector<unsigned> &seedIndexList, const uint8_t *bcl, unsigned nClusters, unsigned tile, const ClusterInfo &complete,
                           unsigned nThreads, std::vector<Seed> &seeds)
{
    std::vector<std::vector<Seed> > parts(nThreads);
    std::vector<std::thread> threads;
    const unsigned clusterLength = p.clusterLength();
    for (unsigned t = 0; t < nThreads; ++t)
        threads.emplace_back([&, t]()
        {
            const uint64_t begin = uint64_t(nClusters) * t / nThreads, end = uint64_t(nClusters) * (t + 1) / nThreads;
            ClusterInfo part(complete.begin() + begin, complete.begin() + end);
            generateSeeds(p, seedIndexList, bcl + begin * clusterLength, unsigned(end - begin), tile, part, parts[t]);
            for (size_t i = 0; i < parts[t].size(); ++i) parts[t][i].seedId += begin << 9;       // SeedId.hh: cluster field at bit 9
        });
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    // (kmer, seed index) leaves seeds of different clusters with equal keys in unspecified order, as the reference's sort does
    for (unsigned width = 1; width < nThreads; width *= 2)
    {
        threads.clear();
        for (unsigned t = 0; t + width < nThreads; t += 2 * width)
            threads.emplace_back([&, t]()
            {
                std::vector<Seed> merged(parts[t].size() + parts[t + width].size());
                std::merge(parts[t].begin(), parts[t].end(), parts[t + width].begin(), parts[t + width].end(), merged.begin(),
                           [](const Seed &l, const Seed &r) { return l.kmer < r.kmer || (l.kmer == r.kmer && SeedId(l.seedId).getSeed() < SeedId(r.seedId).getSeed()); });
                parts[t].swap(merged); std::vector<Seed>().swap(parts[t + width]);
            });
        for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    }
    seeds.swap(parts[0]);
}

void findMatchesExactParallel(const Params &p, const SortedReference &ref, const std::vector<Seed> &seeds, bool closeRepeats, bool storeNoMatches, unsigned nThreads,
                              ClusterInfo &complete, std::vector<Match> &out, std::vector<uint8_t> &contigHasMatches)
{
    const ReferencePosition noMatch(ReferencePosition::NoMatch);
    size_t endSeeds = seeds.size();
    while (endSeeds && SeedId(seeds[endSeeds - 1].seedId).isNSeedId()) --endSeeds;
    // thread t takes the seeds of k-mer range t: cut points moved forward to the next change of k-mer
    std::vector<size_t> cut(nThreads + 1, endSeeds);
    cut[0] = 0;
    for (unsigned t = 1; t < nThreads; ++t)
    {
        size_t at = std::max(cut[t - 1], endSeeds * t / nThreads);
        while (at < endSeeds && at && seeds[at].kmer == seeds[at - 1].kmer) ++at;
        cut[t] = at;
    }
    std::vector<PartialFind> parts(nThreads);
    std::vector<std::string> errors(nThreads);
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < nThreads; ++t)
        threads.emplace_back([&, t]()
        {
            try { parts[t].hits.assign(contigHasMatches.size(), 0); joinRange(p, ref, seeds, cut[t], cut[t + 1], closeRepeats, storeNoMatches, parts[t]); }
            catch (const std::exception &e) { errors[t] = e.what(); }
        });
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    for (unsigned t = 0; t < nThreads; ++t)
    {
        if (!errors[t].empty()) throw std::runtime_error(errors[t]);
        out.insert(out.end(), parts[t].matches.begin(), parts[t].matches.end());
        for (size_t i = 0; i < parts[t].marks.size(); ++i) complete[parts[t].marks[i].first].markReadComplete(parts[t].marks[i].second);
        for (size_t i = 0; i < contigHasMatches.size(); ++i) contigHasMatches[i] |= parts[t].hits[i];
    }
    for (unsigned t = 0; t < nThreads; ++t)
        for (size_t i = 0; i < parts[t].pendingNoMatch.size(); ++i)
        {
            const SeedId id(parts[t].pendingNoMatch[i].seedId);
            if (!complete[id.getCluster()].isReadComplete(p.seeds[id.getSeed()].readIndex)) { Match m = { id.value, noMatch.value }; out.push_back(m); }
        }
    if (storeNoMatches) for (size_t s = endSeeds; s < seeds.size(); ++s) { Match m = { seeds[s].seedId, noMatch.value }; out.push_back(m); }
}
} // namespace

void findTileMatchesParallel(const Params &p, const SortedReference &ref, const uint8_t *bcl, unsigned nClusters, unsigned tile, unsigned nThreads,
                             std::vector<Match> &matches, std::vector<uint8_t> &contigHasMatches)
{
    if (nThreads < 2 || nClusters < nThreads) { findTileMatches(p, ref, bcl, nClusters, tile, matches, contigHasMatches); return; }
    const std::vector<std::vector<unsigned> > perIteration = seedIndexListPerIteration(p.seeds, unsigned(p.reads.size()), p.firstPassSeeds);
    ClusterInfo complete(nClusters);
    matches.clear();
    std::vector<Seed> seeds;
    generateSeedsParallel(p, perIteration.at(0), bcl, nClusters, tile, complete, nThreads, seeds);
    findMatchesExactParallel(p, ref, seeds, false, 1 == perIteration.size(), nThreads, complete, matches, contigHasMatches);
    if (2 == perIteration.size())
    {
        generateSeedsParallel(p, perIteration.at(1), bcl, nClusters, tile, complete, nThreads, seeds);
        findMatchesExactParallel(p, ref, seeds, true, true, nThreads, complete, matches, contigHasMatches);
    }
    // the final order (SelectMatchesTransition.cpp:242-254 + the reverse bit) by cluster ranges on the same threads
    std::vector<std::vector<Match> > byRange(nThreads);
    for (size_t i = 0; i < matches.size(); ++i) byRange[size_t(SeedId(matches[i].seedId).getCluster()) * nThreads / nClusters].push_back(matches[i]);
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < nThreads; ++t)
        threads.emplace_back([&, t]()
        {
            std::sort(byRange[t].begin(), byRange[t].end(), [](const Match &l, const Match &r)
            {
                if (sortByTileBarcodeClusterLocation(l, r)) return true;
                if (sortByTileBarcodeClusterLocation(r, l)) return false;
                return (l.seedId & 1) < (r.seedId & 1);
            });
        });
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    matches.clear();
    for (unsigned t = 0; t < nThreads; ++t) matches.insert(matches.end(), byRange[t].begin(), byRange[t].end());
}

// ---------------------------------------------------------------- index builder
// lib/reference/ReferenceSorter.cpp:105-261: forward-strand k-mers are stored; reverse-complement k-mers only take part in the
// repeat count.  A k-mer with more than `repeatThreshold` (fwd+rc) occurrences is stored as a single TooManyMatch entry.
// Neighbor flag (NeighborsFinder.cpp:192-244,395-446): set when another distinct reference k-mer (either strand) exists within
// Hamming distance 1..4; found as the reference finds it (neighbors.cpp).  Small genomes only.

// nThreads > 1: the sorts run on that many threads (the k-mers of the neighbour lists are distinct and the first sort is stable, so the sorted
// sequences do not depend on the algorithm) and findNeighbors works on its stretches concurrently, as the reference's
// findNeighborsParallel does (NeighborsFinder.cpp:286-309).  The result is the same for any thread count.
SortedReference buildSortedReference(const ContigList &contigs, unsigned seedLength, unsigned repeatThreshold, bool annotateNeighbors, unsigned neighborhoodWidth, unsigned nThreads)
{
    if (nThreads > 1) omp_set_num_threads(int(nThreads));
    if (32 != seedLength) throw std::invalid_argument("only 32-mers are supported");
    struct Entry { uint64_t kmer; uint64_t pos; bool fwd; };
    std::vector<Entry> all;
    for (size_t c = 0; c < contigs.size(); ++c)
    {
        const std::vector<char> &s = contigs[c].forward;
        uint64_t forward = 0, reverse = 0; unsigned bad = 32;
        for (size_t position = 0; position < s.size(); ++position)
        {
            if (bad) --bad;
            unsigned v;
            switch (s[position]) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; break; case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break; default: v = 4; }
            if (v >> 2) bad = 32;
            forward <<= 2; forward |= (v & 3);
            reverse >>= 2; reverse |= uint64_t((~v) & 3) << 62;
            if (0 == bad)
            {
                const uint64_t kmerPosition = position + 1 - 32;
                Entry f = { forward, ReferencePosition(c, kmerPosition, false).value, true }; all.push_back(f);
                Entry r = { reverse, ReferencePosition(c, kmerPosition, true).value, false }; all.push_back(r);
            }
        }
    }
    if (nThreads > 1) __gnu_parallel::stable_sort(all.begin(), all.end(), [](const Entry &a, const Entry &b) { return a.kmer < b.kmer; });
    else std::stable_sort(all.begin(), all.end(), [](const Entry &a, const Entry &b) { return a.kmer < b.kmer; });
    // distinct k-mers (both strands) for the neighbor search
    std::vector<uint64_t> distinct;
    for (size_t i = 0; i < all.size(); ++i) if (distinct.empty() || distinct.back() != all[i].kmer) distinct.push_back(all[i].kmer);
    std::vector<uint8_t> hasNeighbor(distinct.size(), 0);
    if (annotateNeighbors && neighborhoodWidth)
    {
        // NeighborsFinder::generateNeighbors (NeighborsFinder.cpp:192-244): the distinct k-mers of both strands go through the 70
        // block orders of getPermutateList(4); under each they are sorted and compared inside the blocks of equal prefix
        if (4 != neighborhoodWidth) throw std::invalid_argument("the neighbour search is defined for 4 mismatches (NeighborsFinder.cpp:196,343-383)");
        std::vector<AnnotatedKmer<uint64_t> > kmerList(distinct.size());
        for (size_t i = 0; i < distinct.size(); ++i) { kmerList[i].value = distinct[i]; kmerList[i].hasNeighbors = false; }
        const std::vector<Permutate> permutateList = getPermutateList(32, neighborhoodWidth);
        for (size_t k = 0; k < permutateList.size(); ++k)
        {
            const Permutate &permutate = permutateList[k];
            #pragma omp parallel for if (nThreads > 1) schedule(static)
            for (size_t i = 0; i < kmerList.size(); ++i) kmerList[i].value = permutate(kmerList[i].value);
            if (nThreads > 1) __gnu_parallel::sort(kmerList.begin(), kmerList.end()); else std::sort(kmerList.begin(), kmerList.end());
            findNeighbors(kmerList, nThreads > 1 ? 4 * nThreads : 1, nThreads);
        }
        for (size_t i = 0; i < kmerList.size(); ++i) kmerList[i].value = permutateList.back().reorder(kmerList[i].value);
        if (nThreads > 1) __gnu_parallel::sort(kmerList.begin(), kmerList.end()); else std::sort(kmerList.begin(), kmerList.end());
        for (size_t i = 0; i < kmerList.size(); ++i)
        {
            if (kmerList[i].value != distinct[i]) throw std::logic_error("neighbour list out of step with the k-mer list");
            hasNeighbor[i] = kmerList[i].hasNeighbors;
        }
    }
    SortedReference ret;
    for (size_t c = 0; c < contigs.size(); ++c) ret.karyotype.push_back(unsigned(c));
    size_t d = 0;
    for (size_t i = 0; i < all.size();)
    {
        size_t j = i; while (j < all.size() && all[j].kmer == all[i].kmer) ++j;
        while (distinct[d] != all[i].kmer) ++d;
        bool anyFwd = false; for (size_t a = i; a < j; ++a) anyFwd |= all[a].fwd;
        if (anyFwd)
        {
            if (repeatThreshold < j - i)
            {
                ReferenceKmer rk = { all[i].kmer, ReferencePosition(ReferencePosition::TooManyMatch).value }; ret.kmers.push_back(rk);
            }
            else for (size_t a = i; a < j; ++a) if (all[a].fwd)
            {
                ReferenceKmer rk = { all[a].kmer, ReferencePosition::fromValue(all[a].pos).setNeighbors(hasNeighbor[d]).value }; ret.kmers.push_back(rk);
            }
        }
        i = j;
    }
    return ret;
}

void setLookupMode(int mode) { g_lookupMode = mode; }

} // namespace oracle
