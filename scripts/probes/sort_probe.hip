// which hipcub radix sort variants sort (u64 key, u8 value) pairs correctly on this ROCm
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned long long u64; typedef unsigned char u8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename N> int run(size_t n, bool dbl, int beginBit, const char *name, int endBit = 64)
{
    std::vector<u64> h(n); std::vector<u8> hv(n);
    std::mt19937_64 rng(1); for (size_t i = 0; i < n; ++i) { h[i] = rng(); hv[i] = u8(h[i] * 31 >> 7); }
    u64 *a, *b; u8 *va, *vb; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&va, n)); CK(hipMalloc(&vb, n));
    CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(va, hv.data(), n, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    size_t bytes = 0; void *tmp = nullptr; u64 *outK; u8 *outV;
    if (dbl)
    {
        hipcub::DoubleBuffer<u64> dk(a, b); hipcub::DoubleBuffer<u8> dv(va, vb);
        CK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, N(n), beginBit, endBit, st));
        CK(hipMalloc(&tmp, bytes + 16));
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, dk, dv, N(n), beginBit, endBit, st));
        outK = dk.Current(); outV = dv.Current();
    }
    else
    {
        CK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, a, b, va, vb, N(n), beginBit, endBit, st));
        CK(hipMalloc(&tmp, bytes + 16));
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, a, b, va, vb, N(n), beginBit, endBit, st));
        outK = b; outV = vb;
    }
    CK(hipStreamSynchronize(st));
    std::vector<u64> r(n); std::vector<u8> rv(n);
    CK(hipMemcpy(r.data(), outK, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rv.data(), outV, n, hipMemcpyDeviceToHost));
    size_t disorder = 0, badValue = 0;
    for (size_t i = 0; i < n; ++i) { { const u64 m = endBit == 64 ? ~0ull : ((1ull << endBit) - 1); if (i && ((r[i] & m) >> beginBit) < ((r[i - 1] & m) >> beginBit)) ++disorder; } if (rv[i] != u8(r[i] * 31 >> 7)) ++badValue; }
    std::sort(h.begin(), h.end()); std::vector<u64> rs(r); std::sort(rs.begin(), rs.end());
    printf("%-40s n %zu temp %zu: out of order %zu, wrong payload %zu, multiset %s\n", name, n, bytes, disorder, badValue, rs == h ? "kept" : "CHANGED");
    hipFree(a); hipFree(b); hipFree(va); hipFree(vb); hipFree(tmp); hipStreamDestroy(st);
    return 0;
}
int main()
{
    for (size_t n : { size_t(1), size_t(1000), size_t(70000), size_t(700000), size_t(3000000), size_t(20000000) })
    {
        run<size_t>(n, true, 0, "size_t, double buffer, bits 0-32", 32);
        run<int>(n, false, 0, "int, plain, bits 0-58", 58);
        continue;
        run<int>(n, false, 0, "int, plain, bits 0-64");
        run<int>(n, true, 0, "int, double buffer, bits 0-64");
        run<int>(n, true, 32, "int, double buffer, bits 32-64");
        run<size_t>(n, false, 0, "size_t, plain, bits 0-64");
        run<size_t>(n, true, 0, "size_t, double buffer, bits 0-64");
        run<size_t>(n, true, 32, "size_t, double buffer, bits 32-64");
        run<size_t>(n, false, 32, "size_t, plain, bits 32-64");
    }
    return 0;
}
