// The ceiling of the seed lookup: how many random 64-byte lines per second MI355X delivers out of a table-sized buffer (51 GB: 46.9 GB of 16-byte entries + the
// 4.3 GB prefix directory), for the access shapes k_find_matches has or could have.  Every kernel makes N independent line reads, HIP-event timed:
//   one16        a 16-byte load per random line and lane (a probe that wants one entry of a line)
//   one16_x4     four independent random lines per lane in flight before any is used (k_find_matches fetches a slice's entries that way)
//   whole64      a random line read whole by four neighbouring lanes (16 bytes each)
//   sorted16     the same 16-byte loads with the lines in ascending order within a wave's batch (what looking the probes up in k-mer order would give)
//   chained      a 4-byte load from a random line of a 4.3 GB directory, then a 16-byte load from the table line it names: the two dependent round
//                trips of a lookup (directory entry -> table slice)
// Build: hipcc -O3 --offload-arch=gfx950 -o random_lines random_lines.hip; prints lines/s and GB/s in lines (64 B) per kernel.  scripts/exp_r6_random_lines.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_one16(const uint4 *table, uint64_t nLines, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads; i += uint64_t(gridDim.x) * blockDim.x)
    { const uint4 v = table[(mix(i) % nLines) * 4 + (i & 3)]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_one16_x4(const uint4 *table, uint64_t nLines, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < nReads; i += uint64_t(gridDim.x) * blockDim.x * 4)
    {
        const uint4 a = table[(mix(i) % nLines) * 4], b = table[(mix(i + 1) % nLines) * 4 + 1], c = table[(mix(i + 2) % nLines) * 4 + 2], d = table[(mix(i + 3) % nLines) * 4 + 3];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_whole64(const uint4 *table, uint64_t nLines, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads * 4; i += uint64_t(gridDim.x) * blockDim.x)
    { const uint4 v = table[(mix(i >> 2) % nLines) * 4 + (i & 3)]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// lane l of a wave's batch b reads line  (b's random base + l x stride): 64 ascending lines spread over the whole table -- the order a radix sort of the
// probes by their k-mers would give each wave (neighbouring probes 51 GB / 16 M apart)
__global__ void k_sorted16(const uint4 *table, uint64_t nLines, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint64_t nThreads = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads; i += nThreads)
    {
        // the i-th smallest of nReads probes spread evenly with jitter: position ~ i / nReads of the table; consecutive lanes get consecutive ranks
        const uint64_t rank = (i % nThreads) * (nReads / nThreads) + i / nThreads;
        const uint64_t line = (rank * (nLines / nReads)) + mix(i) % (nLines / nReads ? nLines / nReads : 1);
        const uint4 v = table[(line % nLines) * 4 + (i & 3)]; acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_chained(const uint32_t *directory, uint64_t nDirectory, const uint4 *table, uint64_t nLines, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads; i += uint64_t(gridDim.x) * blockDim.x)
    {
        const uint32_t d = directory[mix(i) % nDirectory];                       // (the directory holds random line numbers)
        const uint4 v = table[(uint64_t(d) % nLines) * 4 + (i & 3)]; acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_fill_directory(uint32_t *directory, uint64_t n) { for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) directory[i] = uint32_t(mix(i * 7 + 1)); }

int main(int argc, char **argv)
{
    const uint64_t tableBytes = (argc > 1 ? uint64_t(atoll(argv[1])) : uint64_t(47)) << 30, directoryBytes = uint64_t(43) << 27 /* 4.3 GiB-ish */;
    void *table; uint32_t *directory, *sink;
    CHECK(hipMalloc(&table, tableBytes)); CHECK(hipMalloc((void **)&directory, directoryBytes)); CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(table, 1, tableBytes));
    k_fill_directory<<<4096, 256>>>(directory, directoryBytes / 4);
    const uint64_t nLines = tableBytes / 64, nReads = uint64_t(1) << 27;      // 134 M line reads per kernel (a lookup launch of 1 M pairs makes 33 M)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipDeviceSynchronize());
    const int grids[] = { 2048, 8192, 32768 };
    for (int grid : grids)
        for (int which = 0; which < 5; ++which)
        {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep)
            {
                CHECK(hipEventRecord(e0));
                switch (which)
                {
                case 0: k_one16<<<grid, 256>>>(static_cast<const uint4 *>(table), nLines, nReads, sink); break;
                case 1: k_one16_x4<<<grid, 256>>>(static_cast<const uint4 *>(table), nLines, nReads, sink); break;
                case 2: k_whole64<<<grid, 256>>>(static_cast<const uint4 *>(table), nLines, nReads, sink); break;
                case 3: k_sorted16<<<grid, 256>>>(static_cast<const uint4 *>(table), nLines, nReads, sink); break;
                case 4: k_chained<<<grid, 256>>>(directory, directoryBytes / 4, static_cast<const uint4 *>(table), nLines, nReads, sink); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            static const char *names[] = { "one16", "one16_x4", "whole64", "sorted16", "chained" };
            const double lines = double(nReads) * (4 == which ? 2.0 : 1.0);
            printf("grid %5d %-9s %8.3f ms  %7.2f G lines/s  %7.1f GB/s in 64-byte lines  (%5.1f GB/s of the bytes asked for)\n", grid, names[which], best, lines / best / 1e6, lines * 64 / best / 1e6,
                   double(nReads) * (2 == which ? 64.0 : 4 == which ? 20.0 : 16.0) / best / 1e6);
        }
    return 0;
}
