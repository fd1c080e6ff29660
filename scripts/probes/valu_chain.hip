// How fast does ONE wavefront issue vector instructions on gfx950?  Shader-clock ticks per repetition for
//   0 v_add_u32 (dependent chain)     1 v_pk_add_u16 (dependent)        2 v_pk_max_i16 + v_pk_sub_i16 clamp (dependent pair)
//   3 s_nop 1 + v_mov_b32_dpp row_shr:2 + v_alignbit (dependent)       4 s_nop 1 + v_max_i32_dpp row_shl:2 (dependent)
//   5 two independent v_add_u32 chains interleaved                    6 v_add_u32 chain with an s_add_u32 between each
// Sixteen repetitions per asm statement (the compiler puts an s_nop 0 between asm statements), 4096 in all, for workgroups of 1, 2 and 4
// waves on one CU.  Build: hipcc --offload-arch=gfx950 -O3 valu_chain.hip -o valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
#define R256(x) R16(R16(x))
#define S16(x) x x x x x x x x x x x x x x x x
__global__ void k(unsigned long long *out, int *sink)
{
    int a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, s = 1;
    unsigned long long t[8];
    t[0] = __builtin_readcyclecounter();
    R256(asm volatile(S16("v_add_u32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));)
    t[1] = __builtin_readcyclecounter();
    R256(asm volatile(S16("v_pk_add_u16 %0, %0, %1\n\t") : "+v"(a) : "v"(b));)
    t[2] = __builtin_readcyclecounter();
    R256(asm volatile(S16("v_pk_max_i16 %0, %0, %1\n\tv_pk_sub_i16 %0, %0, %1 clamp\n\t") : "+v"(a) : "v"(b));)
    t[3] = __builtin_readcyclecounter();
    R256(asm volatile(S16("s_nop 1\n\tv_mov_b32_dpp %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_alignbit_b32 %0, %0, %1, 16\n\t") : "+v"(a), "+v"(c));)
    t[4] = __builtin_readcyclecounter();
    R256(asm volatile(S16("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a));)
    t[5] = __builtin_readcyclecounter();
    R256(asm volatile(S16("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\t") : "+v"(a), "+v"(c) : "v"(b));)
    t[6] = __builtin_readcyclecounter();
    R256(asm volatile(S16("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(a), "+s"(s) : "v"(b));)
    t[7] = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 7; ++i) out[i] = t[i + 1] - t[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + c + s;
}
int main()
{
    unsigned long long *out; int *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 1 << 22);
    const char *names[7] = { "v_add_u32 chain", "v_pk_add_u16 chain", "v_pk_max_i16 + v_pk_sub_i16 clamp chain (2 instr)", "s_nop 1 + v_mov_dpp + v_alignbit chain (3 instr)",
                             "s_nop 1 + v_max_i32_dpp chain (2 instr)", "two independent v_add_u32 chains (2 instr)", "v_add_u32 + s_add_u32 (2 instr)" };
    for (int waves = 1; waves <= 4; waves *= 2)
    {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, sink); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, sink); hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, out, 56, hipMemcpyDeviceToHost);
        printf("workgroup of %d wave(s), 4096 repetitions each; shader clock ticks per repetition:\n", waves);
        for (int i = 0; i < 7; ++i) printf("  %-55s %8.3f\n", names[i], double(h[i]) / 4096.0);
    }
    return 0;
}
