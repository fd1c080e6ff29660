// Wavefront form of alignment::BandedSmithWaterman::align (lib/alignment/BandedSmithWaterman.cpp:84-462).
//
// 16 lanes per alignment (one lane per diagonal of the band, exactly the 16 int16 lanes of the reference's two SSE
// registers), 4 alignments per wave64, 16 per 256-thread workgroup.  G/E/F live in registers; the F update and the database
// window move between neighbouring lanes with width-16 wave shuffles; the serial 16-step E chain of the reference
// (:246-297) becomes a 4-step max-plus suffix scan (E[k] = max_{j>k}(max(G'[j],F'[j]) - open - (j-k-1)*extend), which is what
// the chain computes for in-range scores); the traceback flags (one byte per lane per row) are staged in LDS and walked by
// lane 0 of the group.  Integer DP: no MFMA.
#pragma once
#include "types.h"
#include "../../include/isaac_gpu.h"

namespace isaac
{

#if defined(__HIPCC__)

__device__ inline int s16(int v) { return int(short(v)); }

// LDS bytes per alignment group
__host__ __device__ inline u32 bswGroupLdsBytes(u32 maxQueryLength) { return ((maxQueryLength * 16 + 15) & ~15u) + 128; }

__global__ __launch_bounds__(256) void k_bsw_batch(int matchScore, int mismatchScore, int gapOpenScore, int gapExtendScore,
                                                  const char *sequences, const isaac_bsw_job *jobs, u32 nJobs, u32 maxQueryLength,
                                                  isaac_bsw_result *results)
{
    extern __shared__ __align__(16) u8 lds[];
    const u32 group = threadIdx.x >> 4, k = threadIdx.x & 15;
    const u32 job = blockIdx.x * 16 + group;
    if (job >= nJobs) return;
    u8 *T = lds + group * bswGroupLdsBytes(maxQueryLength);
    short *endVals = reinterpret_cast<short *>(T + ((maxQueryLength * 16 + 15) & ~15u));
    const isaac_bsw_job jb = jobs[job];
    const u32 L = jb.query_length;
    const char *query = sequences + jb.query_offset;
    const char *database = sequences + jb.database_offset;
    const int initialValue = s16(-32768 + gapOpenScore);
    const int open = gapOpenScore, ext = gapExtendScore;
    const int wMatch = matchScore & 0xff, wMismatch = s16(0xff00 | (mismatchScore & 0xff));
    int G = (k == 0) ? 0 : initialValue, E = initialValue, F = 0;
    char d = database[15 - k];                      // lane k of row i looks at database[i + 15 - k]
    for (u32 i = 0; i < L; ++i)
    {
        // F: lane k from lane k-1 of the previous row (:130-173)
        const int gp = __shfl_up(G, 1, 16), ep = __shfl_up(E, 1, 16), fp = __shfl_up(F, 1, 16);
        int tf = (gp < ep) ? 1 : 0;
        const int v = s16(max(gp, ep) - open), fe = s16(fp - ext);
        if (v < fe) tf = 2;
        int newF = max(v, fe);
        if (k == 0) { newF = initialValue; tf = 0; }
        // G (:174-197) with the 16-bit max over byte pairs of the flag vectors
        const int fE = (G < E) ? 1 : 0;
        const int m = max(G, E);
        const int fF = (m < F) ? 1 : 0;
        int newG = max(m, F);
        const int pfE = __shfl_xor(fE, 1, 16), pfF = __shfl_xor(fF, 1, 16);
        int tg;
        if (k & 1) tg = fF ? 2 : fE;
        else tg = pfF ? 2 * fF : (pfE ? fE : max(2 * fF, fE));
        // W (:200-244): byte compare, so read 'n' never equals reference 'N'
        const char q = query[i];
        newG = s16(newG + ((q != d) ? wMismatch : wMatch));
        // E (:246-297) as an exclusive max-plus suffix scan over the lanes
        const int g = s16(newG - open), f = s16(newF - open);
        const int NEG = -(1 << 28);
        int c = max(g, f) - int(k) * ext;
        int s = __shfl_down(c, 1, 16); if (k + 1 > 15) s = NEG;
        int t;
        t = __shfl_down(s, 1, 16); if (k + 1 <= 15) s = max(s, t);
        t = __shfl_down(s, 2, 16); if (k + 2 <= 15) s = max(s, t);
        t = __shfl_down(s, 4, 16); if (k + 4 <= 15) s = max(s, t);
        t = __shfl_down(s, 8, 16); if (k + 8 <= 15) s = max(s, t);
        const int newE = (k == 15) ? initialValue : s16(s + int(k + 1) * ext);
        // TE from lane k+1's (g, E - ext, f) with the reference's tie rules
        const int g1 = __shfl_down(g, 1, 16), f1 = __shfl_down(f, 1, 16), e1 = s16(__shfl_down(newE, 1, 16) - ext);
        int te = 0;
        if (k < 15) { if (e1 > g1 && e1 > f1) te = 1; else if (f1 > g1) te = 2; }
        T[i * 16 + k] = u8(tg | (te << 2) | (tf << 4));
        G = newG; E = newE; F = newF;
        // slide the database window: lane k takes lane k-1's base, lane 0 loads the next one
        const char dn = char(__shfl_up(int(d), 1, 16));
        d = (k == 0) ? ((i + 1 < L) ? database[i + 16] : char(0)) : dn;
    }
    endVals[k] = short(G); endVals[16 + k] = short(E); endVals[32 + k] = short(F);
    __syncthreads();   // flags and end values written by the 16 lanes become visible to lane 0 of the group
    if (k != 0) return;
    // end-cell scan (:349-379), traceback (:381-435), stripping of the terminal deletions (:437-453)
    isaac_bsw_result &res = results[job];
    int mx = s16(int(u16(endVals[15])) - 1);
    int ii = int(L) - 1, jj = ii; u32 maxType = 0;
    for (int lane = 15; lane >= 0; --lane)
        for (u32 type = 0; type < 3; ++type)
        {
            const int value = endVals[16 * type + lane];
            if (value > mx) { mx = value; jj = lane; maxType = type; }
        }
    u32 n = 0, opLength = 0;
    u32 *cig = res.cigar;
    bool overflow = false;
#define ISAAC_BSW_PUSH(len, op) do { if (n < ISAAC_GPU_MAX_CIGAR_OPS) cig[n++] = cigarOp(u32(len), op); else overflow = true; } while (0)
    if (jj > 0) ISAAC_BSW_PUSH(jj, OP_DELETE);
    while (ii >= 0 && jj >= 0 && jj <= 15)
    {
        ++opLength;
        const u32 nextMaxType = (T[ii * 16 + jj] >> (2 * maxType)) & 3;
        if (nextMaxType != maxType) { ISAAC_BSW_PUSH(opLength, maxType == 0 ? OP_ALIGN : maxType == 1 ? OP_DELETE : OP_INSERT); opLength = 0; }
        ii += (maxType == 1) ? 0 : -1;
        jj += (maxType == 1) ? 1 : (maxType == 2) ? -1 : 0;
        maxType = nextMaxType;
    }
    if (1 != maxType && opLength) { ISAAC_BSW_PUSH(opLength, maxType == 0 ? OP_ALIGN : OP_INSERT); opLength = 0; }
    if (15 > jj) { ISAAC_BSW_PUSH(opLength + 15 - u32(jj), OP_DELETE); opLength = 0; }
#undef ISAAC_BSW_PUSH
    u32 ret = 0;
    if (n && OP_DELETE == cigarCode(cig[n - 1])) { ret = cigarLen(cig[n - 1]); --n; }
    for (u32 lo = 0, hi = n; lo + 1 < hi; ++lo) { --hi; const u32 tt = cig[lo]; cig[lo] = cig[hi]; cig[hi] = tt; }
    if (n && OP_DELETE == cigarCode(cig[n - 1])) --n;
    res.n_ops = overflow ? 0xffffffffu : n;
    res.offset = ret;
}

#endif // __HIPCC__

} // namespace isaac
