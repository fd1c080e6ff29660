// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
//
// The neighbour annotation of the sorted reference: oligo::Permutate / getPermutateList (lib/oligo/Permutate.cpp:32-170) and
// reference::NeighborsFinder::{generateNeighbors, findNeighbors, findNeighborsParallel, markNeighbors}
// (lib/reference/NeighborsFinder.cpp:192-383).  Pinned by the literals of lib/oligo/cppunit/testPermutate.cpp and
// lib/reference/cppunit/testNeighborsFinder.cpp (tests/golden/oligo.json).
#include "oracle.hpp"

#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <thread>

namespace oracle
{

// Permutate.cpp:59-95: four bits per block; the block at position `origin` of `from` goes to the position it has in `to`
uint64_t Permutate::encode(const std::vector<unsigned> &from, const std::vector<unsigned> &to)
{
    if (from.size() != to.size() || from.size() > 16) throw std::invalid_argument("Permutate: block lists of different sizes");
    uint64_t ret = 0;
    for (size_t origin = 0; origin < from.size(); ++origin)
    {
        const std::vector<unsigned>::const_iterator found = std::find(to.begin(), to.end(), from[origin]);
        if (to.end() == found) throw std::invalid_argument("Permutate: not a permutation");
        ret = (ret << 4) | uint64_t(found - to.begin());
    }
    return ret;
}

Permutate::Permutate(unsigned blockLength, const std::vector<unsigned> &from, const std::vector<unsigned> &to)
    : blockLength_(blockLength), count_(unsigned(from.size())), order_(encode(from, to)), absoluteReverseOrder_(0)
{
    std::vector<unsigned> natural(to.size());
    for (size_t i = 0; i < natural.size(); ++i) natural[i] = unsigned(i);
    absoluteReverseOrder_ = encode(to, natural);
}

// Permutate.cpp:97-122
static void buildPermutationList(const std::vector<unsigned> &prefix, const std::vector<unsigned> &suffix, unsigned n, std::vector<std::vector<unsigned> > &out)
{
    if (prefix.size() == n)
    {
        out.push_back(prefix);
        out.back().insert(out.back().end(), suffix.begin(), suffix.end());
        return;
    }
    for (size_t i = 0; i < suffix.size(); ++i)
        if (prefix.empty() || suffix[i] > prefix.back())
        {
            std::vector<unsigned> newPrefix(prefix), newSuffix(suffix);
            newPrefix.push_back(suffix[i]);
            newSuffix.erase(newSuffix.begin() + i);
            buildPermutationList(newPrefix, newSuffix, n, out);
        }
}

std::vector<Permutate> getPermutateList(unsigned kmerBases, unsigned errorCount)
{
    const unsigned blocksCount = 2 * errorCount;
    if (!errorCount || kmerBases % blocksCount) throw std::invalid_argument("getPermutateList: the k-mer does not divide into 2 * errorCount blocks");
    const unsigned blockLength = kmerBases / blocksCount;
    std::vector<unsigned> suffix(blocksCount);
    for (unsigned i = 0; i < blocksCount; ++i) suffix[i] = i;
    std::vector<std::vector<unsigned> > orders;
    buildPermutationList(std::vector<unsigned>(), suffix, errorCount, orders);
    std::vector<Permutate> ret;
    const std::vector<unsigned> *from = &orders.front();
    for (size_t to = 0; to < orders.size(); ++to) { ret.push_back(Permutate(blockLength, *from, orders[to])); from = &orders[to]; }
    return ret;
}

// NeighborsFinder.cpp:343-383: *blockBegin against everything up to blockEnd
template <typename KmerT>
static void markNeighbors(typename std::vector<AnnotatedKmer<KmerT> >::iterator blockBegin, typename std::vector<AnnotatedKmer<KmerT> >::iterator blockEnd)
{
    const unsigned kmerBases = sizeof(KmerT) * 4;
    AnnotatedKmer<KmerT> &kmer = *blockBegin;
    for (typename std::vector<AnnotatedKmer<KmerT> >::iterator current = blockBegin; blockEnd != current; ++current)
    {
        if (kmer.hasNeighbors && current->hasNeighbors) continue;
        KmerT a = kmer.value, b = current->value;
        unsigned width = kmerBases / 2, mismatchCount = 0;
        while (4 >= mismatchCount && width--)
        {
            const KmerT x = a ^ b;
            if (!x) break;                       // nothing left to differ: equal k-mers are not neighbours
            if (3 & x) ++mismatchCount;
            a >>= 2; b >>= 2;
        }
        if (mismatchCount && 4 >= mismatchCount) { kmer.hasNeighbors = true; current->hasNeighbors = true; }
    }
}

// NeighborsFinder.cpp:311-335
template <typename KmerT>
static void findNeighborsRange(typename std::vector<AnnotatedKmer<KmerT> >::iterator begin, typename std::vector<AnnotatedKmer<KmerT> >::iterator end)
{
    const unsigned prefixShift = sizeof(KmerT) * 4;      // KMER_BASES bits = the upper half of the bases
    while (end != begin)
    {
        const KmerT prefix = begin->value >> prefixShift;
        typename std::vector<AnnotatedKmer<KmerT> >::iterator blockEnd = begin;
        while (end != blockEnd && prefix == (blockEnd->value >> prefixShift)) ++blockEnd;
        for (typename std::vector<AnnotatedKmer<KmerT> >::iterator i = begin; blockEnd > i; ++i) markNeighbors<KmerT>(i, blockEnd);
        begin = blockEnd;
    }
}

// NeighborsFinder.cpp:286-309: the list is cut into `jobs` stretches ending on prefix boundaries (one thread each in the
// reference).  The stretches are independent: they are done one after another, or by `nThreads` threads taking them in turn.
template <typename KmerT>
void findNeighbors(std::vector<AnnotatedKmer<KmerT> > &kmerList, unsigned jobs, unsigned nThreads)
{
    typedef typename std::vector<AnnotatedKmer<KmerT> >::iterator It;
    std::vector<std::pair<It, It> > stretches;
    const unsigned prefixShift = sizeof(KmerT) * 4;
    typename std::vector<AnnotatedKmer<KmerT> >::iterator begin = kmerList.begin();
    unsigned started = 0;
    while (kmerList.end() != begin)
    {
        const unsigned remaining = jobs > started ? jobs - started : 1;
        typename std::vector<AnnotatedKmer<KmerT> >::iterator end = begin + (kmerList.end() - begin) / remaining;
        if (kmerList.end() != end)
        {
            const KmerT prefix = end->value >> prefixShift;
            while (kmerList.end() != end && prefix == (end->value >> prefixShift)) ++end;
        }
        stretches.push_back(std::make_pair(begin, end));
        begin = end; ++started;
    }
    if (nThreads <= 1) { for (size_t i = 0; i < stretches.size(); ++i) findNeighborsRange<KmerT>(stretches[i].first, stretches[i].second); return; }
    std::atomic<size_t> next(0);
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < nThreads; ++t)
        threads.emplace_back([&]() { for (size_t i = next++; i < stretches.size(); i = next++) findNeighborsRange<KmerT>(stretches[i].first, stretches[i].second); });
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
}

template void findNeighbors<uint32_t>(std::vector<AnnotatedKmer<uint32_t> > &, unsigned, unsigned);
template void findNeighbors<uint64_t>(std::vector<AnnotatedKmer<uint64_t> > &, unsigned, unsigned);
template void findNeighbors<unsigned __int128>(std::vector<AnnotatedKmer<unsigned __int128> > &, unsigned, unsigned);

} // namespace oracle
