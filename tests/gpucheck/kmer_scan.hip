// TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product.
//
// An independent witness for the sorted-reference index at sizes no CPU builder reaches (tests/test_gpu_scale.py): for a list of
// query 32-mers the whole reference is scanned by brute force -- every forward 32-mer of every contig against every query -- and
// three facts come back per query, none of which looks at the table under test or shares a line of code with the index builder:
//   count    forward occurrences of the query in the reference
//   possum   sum of their ReferencePosition values (contig + 1 in bits 41.., position in bits 1..40; ReferencePosition.hh:51-188)
//   near     some forward 32-mer of the reference differs from the query in 1..4 bases (what NeighborsFinder.cpp:343-383 calls a neighbour)
// Occurrences and neighbours on the reverse strand are the same facts about the query's reverse complement, which the test submits
// as a query of its own.  A 32-mer is packed as ReferenceSorter / oligo::Kmer do: first base in the top two bits, A 0 C 1 G 2 T 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace
{
const uint32_t MAX_QUERIES = 4096, MAX_CONTIGS = 1024, BLOCK = 256;

__global__ __launch_bounds__(BLOCK) void k_scan(const char *bases, uint64_t totalBases, const uint64_t *contigOffsets, uint32_t nContigs, const uint64_t *queries, uint32_t nQueries,
                                                unsigned long long *count, unsigned long long *possum, uint32_t *near)
{
    __shared__ uint64_t q[MAX_QUERIES];
    __shared__ uint64_t offsets[MAX_CONTIGS + 1];
    for (uint32_t i = threadIdx.x; i < nQueries; i += BLOCK) q[i] = queries[i];
    for (uint32_t i = threadIdx.x; i <= nContigs; i += BLOCK) offsets[i] = contigOffsets[i];
    __syncthreads();
    for (uint64_t p = uint64_t(blockIdx.x) * BLOCK + threadIdx.x; p + 32 <= totalBases; p += uint64_t(gridDim.x) * BLOCK)
    {
        uint32_t lo = 0, hi = nContigs;                 // contig of p: offsets[c] <= p < offsets[c + 1]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (offsets[mid] <= p) lo = mid; else hi = mid; }
        if (p + 32 > offsets[lo + 1]) continue;         // the 32-mer would run into the next contig
        uint64_t kmer = 0; bool valid = true;
        for (uint32_t i = 0; i < 32; ++i)
        {
            const char c = bases[p + i];
            uint32_t v = 4;
            if (c == 'A' || c == 'a') v = 0; else if (c == 'C' || c == 'c') v = 1; else if (c == 'G' || c == 'g') v = 2; else if (c == 'T' || c == 't') v = 3;
            valid &= v < 4;
            kmer = (kmer << 2) | (v & 3);
        }
        if (!valid) continue;
        const uint64_t position = ((uint64_t(lo + 1) << 40) | (p - offsets[lo])) << 1;
        for (uint32_t i = 0; i < nQueries; ++i)
        {
            const uint64_t x = kmer ^ q[i];
            if (!x) { atomicAdd(&count[i], 1ull); atomicAdd(&possum[i], (unsigned long long)position); continue; }
            const uint64_t differing = (x | (x >> 1)) & 0x5555555555555555ull;   // one bit per base that differs
            if (__popcll(differing) <= 4 && !near[i]) near[i] = 1;
        }
    }
}
} // namespace

extern "C" int gpucheck_kmer_scan(const char *bases_dev, const uint64_t *contig_offsets_host, uint32_t n_contigs, const uint64_t *queries_host, uint32_t n_queries,
                                  uint64_t *count_out, uint64_t *possum_out, uint8_t *near_out)
{
    if (n_contigs > MAX_CONTIGS) return 1;
    uint64_t *dOffsets = nullptr, *dQueries = nullptr; unsigned long long *dCount = nullptr, *dSum = nullptr; uint32_t *dNear = nullptr;
    if (hipMalloc(&dOffsets, (n_contigs + 1) * 8) || hipMalloc(&dQueries, MAX_QUERIES * 8) || hipMalloc(&dCount, MAX_QUERIES * 8) || hipMalloc(&dSum, MAX_QUERIES * 8) ||
        hipMalloc(&dNear, MAX_QUERIES * 4)) return 2;
    hipMemcpy(dOffsets, contig_offsets_host, (n_contigs + 1) * 8, hipMemcpyHostToDevice);
    const uint64_t total = contig_offsets_host[n_contigs];
    int rc = 0;
    for (uint32_t done = 0; done < n_queries && !rc; done += MAX_QUERIES)
    {
        const uint32_t n = n_queries - done < MAX_QUERIES ? n_queries - done : MAX_QUERIES;
        hipMemcpy(dQueries, queries_host + done, n * 8, hipMemcpyHostToDevice);
        hipMemset(dCount, 0, MAX_QUERIES * 8); hipMemset(dSum, 0, MAX_QUERIES * 8); hipMemset(dNear, 0, MAX_QUERIES * 4);
        k_scan<<<256 * 32, BLOCK>>>(bases_dev, total, dOffsets, n_contigs, dQueries, n, dCount, dSum, dNear);
        if (hipDeviceSynchronize() != hipSuccess) { rc = 3; break; }
        hipMemcpy(count_out + done, dCount, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(possum_out + done, dSum, n * 8, hipMemcpyDeviceToHost);
        uint32_t hostNear[MAX_QUERIES];
        hipMemcpy(hostNear, dNear, n * 4, hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < n; ++i) near_out[done + i] = uint8_t(hostNear[i] != 0);
    }
    hipFree(dOffsets); hipFree(dQueries); hipFree(dCount); hipFree(dSum); hipFree(dNear);
    return rc;
}
