// Wavefront form of alignment::BandedSmithWaterman::align (lib/alignment/BandedSmithWaterman.cpp:84-462).
//
// 16 lanes per alignment (one lane per diagonal of the band, exactly the 16 int16 lanes of the reference's two SSE
// registers), 4 alignments per wave64, 16 per 256-thread workgroup.  G/E/F live in registers; the F update and the database
// window move between neighbouring lanes with width-16 wave shuffles; the serial 16-step E chain of the reference
// (:246-297) becomes a 4-step max-plus suffix scan (E[k] = max_{j>k}(max(G'[j],F'[j]) - open - (j-k-1)*extend), which is what
// the chain computes for in-range scores); the traceback flags (one byte per lane per row) are staged in LDS and walked by
// lane 0 of the group.  Integer DP: no MFMA.
#pragma once
#include "aligner.h"
#include "../../include/isaac_gpu.h"

namespace isaac
{

#if defined(__HIPCC__)

__device__ inline int s16(int v) { return int(short(v)); }

// Lane exchange inside the 16-lane group of one alignment.  The group is one DPP row, so neighbour shifts are register
// operations (row_shr / row_shl / quad_perm) instead of trips through the LDS crossbar (ds_bpermute).
// (bound_ctrl: a lane whose source falls outside the row reads 0 -- every use below overrides that lane's result -- which leaves the move
// without an `old` operand to copy first)
template <int CTRL> __device__ inline int dpp16(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// lane k <- lane k + 1, the last lane of the row <- `outside` (a lane without a source keeps the `old` operand of the DPP move).
// Written as `last ? outside : rowDown<1>(v)` the exchange would sit in the untaken arm of a branch for lane 15 -- and lane 14, reading a
// lane that is switched off, would keep its own value.
// s = max(s, s of lane k + N) in one instruction: the DPP operand of v_max itself (the compiler emits v_mov, v_mov_dpp, v_max and the
// wait states between them).  Lanes without a source keep s.  The wait states a DPP read needs after a vector write are in the string:
// inline assembly is opaque to the hazard recogniser.
template <int N> __device__ inline int maxRowDown(int s)
{
    int r = s;
    asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_shl:%2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(s), "n"(N));
    return r;
}
__device__ inline int rowDown1Or(int v, int outside) { return __builtin_amdgcn_update_dpp(outside, v, 0x101, 0xf, 0xf, false); }
__device__ inline int rowUp1(int v) { return dpp16<0x111>(v); }          // lane k <- lane k - 1   (row_shr:1)
__device__ inline int rowXor1(int v) { return dpp16<0xB1>(v); }          // lane k <- lane k ^ 1   (quad_perm [1,0,3,2])
template <int N> __device__ inline int rowDown(int v) { return dpp16<0x100 + N>(v); }   // lane k <- lane k + N (row_shl:N)

// LDS bytes per alignment group
// bswGroupLdsBytes / gappedGroupLdsBytes: kernels.h (the host sizes the launches with them)

// The DP of one alignment on the 16 lanes of a group, then traceback and CIGAR on lane 0.  `cig[n..)` receives the operations
// (reference order); the return value (lane 0 only) is BandedSmithWaterman::align's: the length of the stripped leading
// deletion.  T: L*16 bytes of LDS, endVals: 48 shorts of LDS, both private to the group.  The 16 lanes are part of one
// wave, so LDS traffic between them needs no workgroup barrier.
// PADDED: query[L] and database[L + 16] may be read (staged copies with room behind them): the look-ahead then needs no clamping
template <bool PADDED = false, typename QueryF>
__device__ inline u32 bswCooperative(int matchScore, int mismatchScore, int gapOpenScore, int gapExtendScore, QueryF query, u32 L, const char *database,
                                     u8 *T, short *endVals, u32 k, u32 *cig, u32 cap, u32 &n, bool &overflow)
{
    STAMP_BEGIN();
    const int initialValue = s16(-32768 + gapOpenScore);
    const int open = gapOpenScore, ext = gapExtendScore;
    const int wMatch = matchScore & 0xff, wMismatch = s16(0xff00 | (mismatchScore & 0xff));
    int G = (k == 0) ? 0 : initialValue, E = initialValue, F = 0;
    const bool first = k == 0, last = k == 15, odd = (k & 1) != 0;
    const int kExt = int(k) * ext, k1Ext = int(k + 1) * ext;
    int d = u8(database[15 - k]);                   // lane k of row i looks at database[i + 15 - k]
    // the next row's query base and the database base that enters the band with it are requested a row ahead (every lane reads the same
    // bytes: one broadcast access, no branch), so that their latency lies behind the row's arithmetic
    int qNext = u8(query(0)), dNext = u8(database[L > 1 ? 16 : 15]);
    for (u32 i = 0; i < L; ++i)
    {
        const int q = qNext, dIn = dNext;
        if (PADDED) { qNext = u8(query(i + 1)); dNext = u8(database[i + 17]); }
        else
        {
            const u32 ahead = i + 1 < L ? i + 1 : i;                  // the values fetched in the last row are not used
            qNext = u8(query(ahead));
            dNext = u8(database[ahead + 1 < L ? ahead + 16 : ahead + 15]);
        }
        // F: lane k from lane k-1 of the previous row (:130-173)
        const int gp = rowUp1(G), ep = rowUp1(E), fp = rowUp1(F);
        const int v = s16(max(gp, ep) - open), fe = s16(fp - ext);
        int tf = (v < fe) ? 2 : ((gp < ep) ? 1 : 0);
        int newF = max(v, fe);
        newF = first ? initialValue : newF; tf = first ? 0 : tf;
        // G (:174-197) with the 16-bit max over byte pairs of the flag vectors
        const int fE = (G < E) ? 1 : 0;
        const int m = max(G, E);
        const int fF = (m < F) ? 1 : 0;
        int newG = max(m, F);
        const int pfE = rowXor1(fE), pfF = rowXor1(fF);
        const int tgOdd = fF ? 2 : fE;
        const int tgEven = pfF ? 2 * fF : (pfE ? fE : max(2 * fF, fE));
        const int tg = odd ? tgOdd : tgEven;
        // W (:200-244): byte compare, so read 'n' never equals reference 'N'
        newG = s16(newG + ((q != d) ? wMismatch : wMatch));
        // E (:246-297) as an exclusive max-plus suffix scan over the lanes.  A lane whose source lies outside the row gets its own value
        // back from the row shift, and max(s, s) = s: only the first step needs to know where the row ends
        const int g = s16(newG - open), f = s16(newF - open);
        const int NEG = -(1 << 28);
        const int c = max(g, f) - kExt;
        int s = rowDown1Or(c, NEG);
        s = maxRowDown<1>(s);
        s = maxRowDown<2>(s);
        s = maxRowDown<4>(s);
        s = maxRowDown<8>(s);
        const int newE = last ? initialValue : s16(s + k1Ext);
        // TE from lane k+1's (g, E - ext, f) with the reference's tie rules
        const int g1 = rowDown<1>(g), f1 = rowDown<1>(f), e1 = s16(rowDown<1>(newE) - ext);
        int te = (e1 > g1 && e1 > f1) ? 1 : ((f1 > g1) ? 2 : 0);
        te = last ? 0 : te;
        T[i * 16 + k] = u8(tg | (te << 2) | (tf << 4));
        G = newG; E = newE; F = newF;
        // slide the database window: lane k takes lane k-1's base, lane 0 the next one (what it takes in the last row is not looked at)
        const int dn = rowUp1(d);
        d = first ? dIn : dn;
    }
    endVals[k] = short(G); endVals[16 + k] = short(E); endVals[32 + k] = short(F);
    STAMP(55);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32 ret = 0;
    {
        // end-cell scan (:349-379), traceback (:381-435), stripping of the terminal deletions (:437-453).  All 16 lanes of the group
        // walk the traceback together (the same values in every lane; lane 0 does the stores): a stretch of ALIGN cells whose flags
        // say "came from ALIGN" -- nearly all of a typical alignment -- is crossed 16 rows at a time, every lane looking at one row,
        // instead of one dependent LDS read per row.
        const u32 first = n;
        int mx = s16(int(u16(endVals[15])) - 1);
        int ii = int(L) - 1, jj = ii; u32 maxType = 0;
        for (int lane = 15; lane >= 0; --lane)
            for (u32 type = 0; type < 3; ++type)
            {
                const int value = endVals[16 * type + lane];
                if (value > mx) { mx = value; jj = lane; maxType = type; }
            }
        u32 opLength = 0, firstOp = 0, lastOp = 0;
        const u32 groupShift = (threadIdx.x & 63u) & ~15u;
#define ISAAC_BSW_PUSH(len, op) do { if (n < cap) { lastOp = cigarOp(u32(len), op); if (n == first) firstOp = lastOp; if (k == 0) cig[n] = lastOp; ++n; } else overflow = true; } while (0)
        if (jj > 0) ISAAC_BSW_PUSH(jj, OP_DELETE);
        while (ii >= 0 && jj >= 0 && jj <= 15)
        {
            if (0 == maxType)
            {
                const int row = ii - int(k);
                const bool stop = row < 0 || 0 != (T[row * 16 + jj] & 3);
                const u32 mask = u32(__ballot(stop) >> groupShift) & 0xffffu;
                const u32 run = mask ? u32(__ffs(int(mask))) - 1 : 16u;
                opLength += run; ii -= int(run);
                if (!mask) continue;
                if (ii < 0) break;
            }
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
        // the operations were pushed back to front: the last one is the alignment's leading deletion, the first one its trailing one
        if (n > first && OP_DELETE == cigarCode(lastOp)) { ret = cigarLen(lastOp); --n; }
        if (k == 0) for (u32 lo = first, hi = n; lo + 1 < hi; ++lo) { --hi; const u32 tt = cig[lo]; cig[lo] = cig[hi]; cig[hi] = tt; }
        if (n > first && OP_DELETE == cigarCode(firstOp)) --n;
    }
    STAMP(56);
    // the group's LDS is reused by the next problem only after lane 0 is done with it
    __builtin_amdgcn_wave_barrier();
    return ret;
}

struct PlainQuery { const char *q; __device__ char operator()(u32 i) const { return q[i]; } };

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
    PlainQuery q; q.q = sequences + jb.query_offset;
    isaac_bsw_result &res = results[job];
    u32 n = 0; bool overflow = false;
    const u32 ret = bswCooperative(matchScore, mismatchScore, gapOpenScore, gapExtendScore, q, jb.query_length, sequences + jb.database_offset, T, endVals, k,
                                   res.cigar, ISAAC_GPU_MAX_CIGAR_OPS, n, overflow);
    if (k == 0) { res.n_ops = overflow ? 0xffffffffu : n; res.offset = ret; }
}

struct StrandQueryDev { ReadView read; bool reverse; u32 offset; __device__ char operator()(u32 i) const { return strandBase(read, reverse, offset + i); } };

// GappedAligner::alignGapped (GappedAligner.cpp:167-249) for a list of candidates, 16 lanes per candidate: the statements of
// alignGapped() in aligner.h with the DP on the group and everything else on its lane 0.  `bcl` is the tile, the job's
// cluster index is relative to clusterBase.  Grid-stride over the jobs, so the launch does not need the job count on the host.
__global__ __launch_bounds__(256) void k_gapped_jobs(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                    u32 maxReadLength, GappedResult *results)
{
    extern __shared__ __align__(16) u8 lds[];
    __shared__ double qualityTables[128];
    for (u32 qi = threadIdx.x; qi < 128; qi += blockDim.x) qualityTables[qi] = qi < 64 ? Rg.logMatch[qi] : Rg.logMismatch[qi - 64];
    __syncthreads();
    DevReference R = Rg; R.logMatch = qualityTables; R.logMismatch = qualityTables + 64;
    const u32 group = threadIdx.x >> 4, k = threadIdx.x & 15;
    u8 *T = lds + group * gappedGroupLdsBytes(maxReadLength);
    short *endVals = reinterpret_cast<short *>(T + ((maxReadLength * 16 + 15) & ~15u));
    char *stagedQuery = reinterpret_cast<char *>(T + bswGroupLdsBytes(maxReadLength));
    char *stagedDatabase = stagedQuery + ((maxReadLength + 31) & ~15u);
    const u32 nJobs = imin(*jobCounter, jobsCap);
    for (u32 j = blockIdx.x * 16 + group; j < nJobs; j += gridDim.x * 16)
    {
        STAMP_BEGIN();
        const GappedJob &jb = jobs[j];
        GappedResult &res = results[j];
        Cand f = jb.in;
        const u32 r = f.readIndex;
        ReadView read;
        read.bcl = bcl + u64(clusterBase + jb.cluster) * P.clusterLength + P.readOffset[r]; read.length = P.readLength[r];
        read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = jb.endCyclesMasked;
        CigarPool pool; pool.words = res.cigar; pool.used = 0; pool.capacity = 40; pool.overflow = 0;
        candResetAlignment(f, pool);
        f.lowClipped = 0; f.highClipped = 0;
        const u64 referenceSize = contigLength(R, f.contigId);
        i64 begin, end;
        bool go = clipSequence(read, f, i64(referenceSize), begin, end);
        u32 n = 0, matchCount = 0; bool overflow = false;
        u32 sequenceLength = 0; i64 strandPosition = 0;
        if (go)
        {
            if (begin) { if (k == 0) res.cigar[0] = cigarOp(u32(begin), OP_SOFT_CLIP); n = 1; }
            sequenceLength = u32(end - begin);
            strandPosition = f.position;
            if (i64(referenceSize) < i64(sequenceLength) + strandPosition + i64(BSW_WIDEST_GAP_SIZE)) go = false;
            if (!sequenceLength || sequenceLength > maxReadLength) go = false;
        }
        if (go)
        {
            u32 left, right;
            getFlanks(strandPosition, sequenceLength, referenceSize, left, right);
            const char *database = R.bases + R.contigOffset[f.contigId] + strandPosition - left;
            STAMP(50);
            // the group's 16 lanes bring the query and the window into LDS side by side; the DP rows then read one byte of each
            for (u32 i = k; i < sequenceLength; i += 16) stagedQuery[i] = strandBase(read, f.reverse, u32(begin) + i);
            for (u32 i = k; i < sequenceLength + 16; i += 16) stagedDatabase[i] = (i < sequenceLength + 15) ? database[i] : char(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            STAMP(51);
            PlainQuery q; q.q = stagedQuery;
            const u32 ret = bswCooperative<true>(P.gapMatch, P.gapMismatch, -P.gapOpen, -P.gapExtend, q, sequenceLength, stagedDatabase, T, endVals, k, res.cigar, 40u, n, overflow);
            STAMP(52);
            if (k == 0)
            {
                strandPosition += ret;
                const u32 clipEndBases = u32(i64(read.length) - end);
                if (clipEndBases) { if (n < 40) res.cigar[n++] = cigarOp(clipEndBases, OP_SOFT_CLIP); else overflow = true; }
                strandPosition -= left;
                // the rescan of the CIGAR (a serial fp64 chain per alignment) is k_gapped_rescan's, one thread per problem:
                // here it would occupy one lane in sixteen.  Handed over: the strand position and "aligned" in matchCount.
                f.position = strandPosition;
                matchCount = 1;
            }
        }
        STAMP(53);
        if (k == 0) { res.out = f; res.matchCount = matchCount; res.nCigar = overflow ? 0xffffffffu : n; }
        STAMP(54);
    }
}

// AlignerBase::updateFragmentCigar for the alignments k_gapped_jobs produced: one thread per problem
__global__ __launch_bounds__(256) void k_gapped_rescan(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                      GappedResult *results)
{
    __shared__ double qualityTables[128];
    for (u32 qi = threadIdx.x; qi < 128; qi += blockDim.x) qualityTables[qi] = qi < 64 ? Rg.logMatch[qi] : Rg.logMismatch[qi - 64];
    __syncthreads();
    DevReference R = Rg; R.logMatch = qualityTables; R.logMismatch = qualityTables + 64;
    const u32 nJobs = imin(*jobCounter, jobsCap);
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < nJobs; j += gridDim.x * blockDim.x)
    {
        GappedResult &res = results[j];
        if (!res.matchCount || 0xffffffffu == res.nCigar) continue;      // refused by the aligner, or the CIGAR did not fit
        const GappedJob &jb = jobs[j];
        Cand f = res.out;
        const u32 r = f.readIndex;
        ReadView read;
        read.bcl = bcl + u64(clusterBase + jb.cluster) * P.clusterLength + P.readOffset[r]; read.length = P.readLength[r];
        read.firstCycle = P.firstCycle[r]; read.endCyclesMasked = jb.endCyclesMasked;
        CigarPool pool; pool.words = res.cigar; pool.used = res.nCigar; pool.capacity = 40; pool.overflow = 0;
        const i64 strandPosition = f.position;
        res.matchCount = updateFragmentCigar(P, R, read, f, strandPosition, pool, 0);
        res.out = f;
    }
}

#endif // __HIPCC__

} // namespace isaac
