// What rocprofv3's FETCH_SIZE reports for the access patterns of this library's kernels, on buffers far larger than the Infinity Cache so that every
// request goes to memory: the guide (MI355X_MICROARCH.md, HBM) calibrates the counter on wide streaming reads only -- it reports half their bytes -- and says
// that other widths are uncalibrated.  Four kernels, each reading a known number of bytes / cache lines:
//   k_stream16    every lane 16 consecutive bytes, lanes consecutive: the guide's pattern
//   k_record64    random 64-byte records (64-byte aligned), four lanes a record, 16 bytes a lane: a candidate record read whole
//   k_line16      random 64-byte lines, one 16-byte load per line and lane: a field of a record
//   k_gather8     random 8-byte words: a table probe
// Build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip; run under rocprofv3 --pmc FETCH_SIZE (scripts/exp_fetch_calib.sh prints the ratios).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void k_stream16(const uint4 *in, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_record64(const uint4 *in, uint64_t nRecords, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads * 4; i += uint64_t(gridDim.x) * blockDim.x)
    { const uint64_t r = mix(i >> 2) % nRecords; const uint4 v = in[r * 4 + (i & 3)]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_line16(const uint4 *in, uint64_t nRecords, uint64_t nReads, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads; i += uint64_t(gridDim.x) * blockDim.x)
    { const uint64_t r = mix(i) % nRecords; const uint4 v = in[r * 4 + 1]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_gather8(const uint64_t *in, uint64_t nWords, uint64_t nReads, uint32_t *sink)
{
    uint64_t acc = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nReads; i += uint64_t(gridDim.x) * blockDim.x) acc ^= in[mix(i) % nWords];
    if (acc == 0x12345678u) *sink = uint32_t(acc);
}

int main()
{
    const uint64_t bytes = uint64_t(8) << 30;            // 8 GB: 32 x the Infinity Cache
    void *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const uint64_t nReads = uint64_t(1) << 26;           // 64 M accesses per kernel
    hipDeviceSynchronize();
    k_stream16<<<8192, 256>>>(static_cast<const uint4 *>(buf), bytes / 16, sink);
    k_record64<<<8192, 256>>>(static_cast<const uint4 *>(buf), bytes / 64, nReads, sink);
    k_line16<<<8192, 256>>>(static_cast<const uint4 *>(buf), bytes / 64, nReads, sink);
    k_gather8<<<8192, 256>>>(static_cast<const uint64_t *>(buf), bytes / 8, nReads, sink);
    hipDeviceSynchronize();
    printf("expected bytes: k_stream16 %llu, k_record64 %llu (records x 64), k_line16 %llu (lines x 64; %llu used), k_gather8 %llu (lines x 64; %llu used)\n",
           (unsigned long long)bytes, (unsigned long long)(nReads * 64), (unsigned long long)(nReads * 64), (unsigned long long)(nReads * 16), (unsigned long long)(nReads * 64), (unsigned long long)(nReads * 8));
    return 0;
}
