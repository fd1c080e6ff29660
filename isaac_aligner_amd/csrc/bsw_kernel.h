// Wavefront form of alignment::BandedSmithWaterman::align (lib/alignment/BandedSmithWaterman.cpp:84-462).
//
// 8 lanes per alignment, two cells of the 16-wide band per lane: lane l holds cells 2l (low 16 bits) and 2l + 1 (high 16 bits) of G, E
// and F in one register each, so that the row's arithmetic is packed int16 (v_pk_add / sub / max: the wrapping _mm_*_epi16 of the
// reference, two cells an instruction) and the reference's "16-bit max over pairs of flag bytes" pairs the two halves of a register.
// The two alignments of a 16-lane DPP row are interleaved (even lanes one, odd lanes the other): a shift by two lanes moves along one
// alignment and never crosses into the other, and the row's ends behave as the band's ends.  8 alignments per wave64, 16 per 128-thread
// workgroup.  A cell's left / right neighbour is the other half of the register or the neighbouring lane's: one DPP move and one
// v_alignbit.  What a cell passes on -- the F it gives to the cell above it in the next row, the E-traceback flag of the cell below --
// is computed where its inputs are and then moved, instead of moving the inputs.  The serial 16-step E chain of the reference
// (:246-297) is an exclusive max-plus suffix scan (E[k] = max_{j>k}(max(G'[j],F'[j]) - open - (j-k-1)*extend)): both cells of a lane in
// 32 bits (the chain's intermediate values are not wrapped), then the maximum over the lanes above (seven DPP reads of one register).  The traceback flags
// (one of 27 codes per cell and row: which of G / E / F each of the cell's three values came from; five bits, ten bytes per row) are staged in LDS
// and walked by the group together.  Integer DP: no MFMA.
// (Round 2's form had one cell per lane, 16 lanes per alignment, 32-bit arithmetic with a sign extension after every operation: ~75
// issue slots per row for 4 alignments; this one ~80 for 8.)
#pragma once
#include "aligner.h"
#include "../../include/isaac_gpu.h"

namespace isaac
{

#if defined(__HIPCC__)

__device__ inline int s16(int v) { return int(short(v)); }

typedef short S2 __attribute__((ext_vector_type(2)));
typedef unsigned short U2 __attribute__((ext_vector_type(2)));
__device__ inline S2 asS2(int v) { return __builtin_bit_cast(S2, v); }
__device__ inline S2 asS2(U2 v) { return __builtin_bit_cast(S2, v); }
__device__ inline U2 asU2(int v) { return __builtin_bit_cast(U2, v); }
__device__ inline int asInt(S2 v) { return __builtin_bit_cast(int, v); }
__device__ inline int asInt(U2 v) { return __builtin_bit_cast(int, v); }
__device__ inline S2 pkMax(S2 a, S2 b) { return __builtin_elementwise_max(a, b); }
__device__ inline U2 pkMaxU(U2 a, U2 b) { return __builtin_elementwise_max(a, b); }
__device__ inline U2 pkMinU(U2 a, U2 b) { return __builtin_elementwise_min(a, b); }
// 1 where a < b, per half: the sign of the saturated difference (a plain difference wraps)
__device__ inline U2 pkLt(S2 a, S2 b) { return __builtin_bit_cast(U2, __builtin_elementwise_sub_sat(a, b)) >> 15; }
__device__ inline int both(int v) { return int((u32(v) & 0xffffu) | (u32(v) << 16)); }
// (mask & a) | (~mask & b): v_bfi_b32
__device__ inline int bfi(int mask, int a, int b) { return (mask & a) | (~mask & b); }
// keeps a packed 0 / 1 flag pair a value of its own: the optimiser otherwise turns arithmetic on it back into a compare and a select per half
__device__ inline U2 opaque(U2 v) { int x = asInt(v); asm("" : "+v"(x)); return asU2(x); }
// 1 where the half is not zero (min(x, 1), which the optimiser would rewrite as a compare and a select per half)
__device__ inline U2 pkNonZero(int x, int ones) { int r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(ones)); return asU2(r); }

// Lane exchange along one alignment: its lanes are every second lane of a DPP row.  A lane without a source (the row's end) keeps `old`.
__device__ inline int fromLaneBelow(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x112, 0xf, 0xf, false); }    // lane l <- lane l - 1 (row_shr:2)
__device__ inline int fromLaneAbove(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, 0x102, 0xf, 0xf, false); }    // lane l <- lane l + 1 (row_shl:2)
// every cell takes the value of the cell below it (k - 1); cell 0 takes the low half of `first`
__device__ inline int fromCellBelow(int v, int first) { return int(__builtin_amdgcn_alignbit(u32(v), u32(fromLaneBelow(v, int(u32(first) << 16))), 16)); }
// every cell takes the value of the cell above it (k + 1); cell 15 takes the low half of `last`
__device__ inline int fromCellAbove(int v, int last) { return int(__builtin_amdgcn_alignbit(u32(fromLaneAbove(v, last)), u32(v), 16)); }
// v_max with the DPP operand in the instruction itself (the compiler emits v_mov, v_mov_dpp, v_max and the wait states between them).
// Lanes without a source keep s.  Inline assembly is opaque to the hazard recogniser: the wait states a DPP read needs after a vector
// write of the same register are the caller's business.
// s = max(s, t of lane l + N / 2): the DPP operand is t, which was written long before, so these need no wait states between them.
template <int N> __device__ inline int maxWithLaneAbove(int s, int t)
{
    asm("v_max_i32_dpp %0, %1, %0 row_shl:%2 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(t), "n"(N));
    return s;
}

// LDS bytes per alignment group
// bswGroupLdsBytes / gappedGroupLdsBytes: kernels.h (the host sizes the launches with them)

// which alignment group of its workgroup a thread belongs to, and which of the group's 8 lanes it is
__device__ inline u32 bswGroupOfThread() { return ((threadIdx.x >> 4) << 1) | (threadIdx.x & 1u); }
__device__ inline u32 bswLaneOfThread() { return (threadIdx.x & 15u) >> 1; }
// BSW_GROUP_LANES, BSW_BLOCK: kernels.h

// The DP of one alignment on the 8 lanes of a group, then traceback and CIGAR (the group walks together, lane 0 stores).  `cig[n..)`
// receives the operations (reference order); the return value is BandedSmithWaterman::align's: the length of the stripped leading
// deletion.  T: bswFlagBytes(L) bytes of LDS, endVals: 48 shorts of LDS, both private to the group.  The 8 lanes are part of one
// wave, so LDS traffic between them needs no workgroup barrier.
// Where the row loop takes its bases from (SOURCE):
// BSW_FROM_FUNCTION: query(i) and database[i], a byte at a time (k_bsw_batch: sequences in device memory)
// BSW_FROM_LDS (round 5's k_gapped_jobs, kept for reads beyond what the registers hold): query.q and database are staged copies in LDS with room behind them
//   (query[L] and database[L + 16] may be read): four rows' bases per 32-bit read, requested ahead
// BSW_FROM_REGISTERS: `query` is a BswRegisterWords -- the group's eight lanes hold the query and the window eight bases per 64-bit register, and a block of
//   eight rows gets its bases by four lane permutes (ds_bpermute: the LDS crossbar, no LDS memory).  `database` is not looked at.
// GLOBAL_FLAGS: T is device memory, not LDS (1.8 KB of flags per problem at 2 x 150 limit the wavefronts a CU holds to two per SIMD, and a row is
// a chain of dependent packed and DPP instructions that would like more of them to hide behind).  The stores are plain, the traceback's reads go to
// the L2 (agent scope) behind a release fence: they are other lanes' stores.  An experiment that lost (kernels.h: ISAAC_BSW_GLOBAL_FLAGS).
struct BswNothingBetween { __device__ void operator()() const {} };
// between(): called once, behind the last row and before the traceback (k_gapped_jobs asks for its next problem's bases there: they arrive while the group walks back)
enum { BSW_FROM_FUNCTION = 0, BSW_FROM_LDS = 1, BSW_FROM_REGISTERS = 2 };
// The sequences of a problem in the registers of its group: chunk c (bases 8c .. 8c + 7; the query as ASCII ACGTn, the window from its first base on) is
// register c / 8 of lane c % 8.  The blocks are asked for in order, so a register is done with after eight blocks and the arrays move down one place:
// the permutes always read element 0 and no register is ever indexed at run time.
template <u32 NCH> struct BswRegisterWords
{
    u64 q[NCH], w[NCH];
    u32 addressBase;         // byte address (lane x 4) of the group's lane 0 for ds_bpermute
    __device__ static u32 permute(u32 address, u32 value) { return u32(__builtin_amdgcn_ds_bpermute(int(address), int(value))); }
    // the query's bases of rows 8b .. 8b + 7 (q0: the first four, a byte each) and the window's bases that enter the band behind them (window[8b + 16 ..])
    __device__ void block(u32 b, u32 &q0, u32 &q1, u32 &d0, u32 &d1)
    {
        if (b && !(b & 7)) { _Pragma("unroll") for (u32 t = 0; t + 1 < NCH; ++t) q[t] = q[t + 1]; }
        if (!((b + 2) & 7)) { _Pragma("unroll") for (u32 t = 0; t + 1 < NCH; ++t) w[t] = w[t + 1]; }
        const u32 fromQ = addressBase | ((b & 7) << 3), fromD = addressBase | (((b + 2) & 7) << 3);
        q0 = permute(fromQ, u32(q[0])); q1 = permute(fromQ, u32(q[0] >> 32));
        d0 = permute(fromD, u32(w[0])); d1 = permute(fromD, u32(w[0] >> 32));
    }
    // window[14 - 2l] in the low byte, window[15 - 2l] above it: what lane l's two cells look at in the first row (before any block() call)
    __device__ u32 firstWindowBases(u32 l) const
    {
        const u32 from = addressBase | ((l < 4 ? 1u : 0u) << 3);             // window[8 .. 15] is lane 1's, window[0 .. 7] lane 0's
        const u32 lo = permute(from, u32(w[0])), hi = permute(from, u32(w[0] >> 32));
        const u32 at = (6 - 2 * l) & 7;                                     // byte of the chunk where window[14 - 2l] lies
        return ((at < 4 ? lo : hi) >> (8 * (at & 3))) & 0xffffu;
    }
};
template <int SOURCE = BSW_FROM_FUNCTION, bool GLOBAL_FLAGS = false, typename QueryF, typename BetweenF = BswNothingBetween>
__device__ inline u32 bswCooperative(int matchScore, int mismatchScore, int gapOpenScore, int gapExtendScore, QueryF &query, u32 L, const char *database,
                                     u8 *T, short *endVals, u32 l, u32 *cig, u32 cap, u32 &n, bool &overflow, BetweenF between = BetweenF())
{
    STAMP_BEGIN();
    const int initialValue = s16(-32768 + gapOpenScore);
    const S2 open2 = asS2(both(gapOpenScore)), ext2 = asS2(both(gapExtendScore));
    const U2 wMatch2 = asU2(both(matchScore & 0xff)), wDiff2 = asU2(both((0xff00 | (mismatchScore & 0xff)) - (matchScore & 0xff)));
    const int init2 = both(initialValue), ones2 = 0x00010001;
    int G = (l == 0) ? int(u32(initialValue) << 16) : init2, E = init2, F = 0;                 // cell 0 starts at 0 (:110-113)
    const bool lastLane = l == 7;
    const int ext = gapExtendScore;
    const int kExtLo = int(2 * l) * ext, kExtHi = int(2 * l + 1) * ext, k1ExtHi = int(2 * l + 2) * ext;        // k1Ext of the low cell = kExtHi
    int d2;                                                                                    // cell k of row i looks at database[i + 15 - k]
    if constexpr (BSW_FROM_REGISTERS == SOURCE) { const u32 two = query.firstWindowBases(l); d2 = int(two >> 8) | (int(two & 0xffu) << 16); }
    else d2 = int(u8(database[15 - 2 * l])) | (int(u8(database[14 - 2 * l])) << 16);
    // Traceback flags: tf + 3 te + 9 tg per cell -- 27 codes, five bits -- ten bits per lane and row, eight rows of a lane in ten bytes: ten bytes per
    // row and alignment (round 5: six bits per cell, twelve bytes; 16 with a byte per cell).  1.5 KB per alignment at 2 x 150, which with the sequences
    // in registers (k_gapped_jobs) lets a CU hold three wavefronts per SIMD instead of two.
    // Block b (rows 8b .. 8b + 7), lane l: 80 bits, row j's ten at bit 10 j (low cell first): the first 64 at T + 80 b + 8 l, the last 16 at T + 80 b + 64 + 2 l.
    const auto storeFlags = [&](u32 block, const u32 (&f)[8])
    {
        const u32 w0 = f[0] | (f[1] << 10) | (f[2] << 20) | (f[3] << 30), w1 = (f[3] >> 2) | (f[4] << 8) | (f[5] << 18) | (f[6] << 28), w2 = (f[6] >> 4) | (f[7] << 6);
        if constexpr (GLOBAL_FLAGS)
        {
            typedef __attribute__((address_space(1))) u32 GlobalU32;
            typedef __attribute__((address_space(1))) u16 GlobalU16;
            GlobalU32 *to = (GlobalU32 *)(reinterpret_cast<u32 *>(T + block * 80 + l * 8));
            to[0] = w0; to[1] = w1; *(GlobalU16 *)(reinterpret_cast<u16 *>(T + block * 80 + 64 + l * 2)) = u16(w2);
        }
        else
        {
            *reinterpret_cast<uint2 *>(T + block * 80 + l * 8) = make_uint2(w0, w1);
            *reinterpret_cast<u16 *>(T + block * 80 + 64 + l * 2) = u16(w2);
        }
    };
    // the code of cell `cell` in row r
    const auto flagsAt = [&](int r, int cell) -> u32
    {
        const u32 bit = u32(r & 7) * 10 + u32(cell & 1) * 5, n = bit >> 3;
        const u8 *block = T + u32(r >> 3) * 80, *first = block + u32(cell >> 1) * 8, *second = block + 64 + u32(cell >> 1) * 2 - 8;
        const u8 *at0 = (n < 8 ? first : second) + n, *at1 = (n < 7 ? first : second) + n + 1;       // (byte 10, past the lane's ten, is read and not looked at)
        u32 b0, b1;
        if constexpr (GLOBAL_FLAGS)
        {
            b0 = __hip_atomic_load(at0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b1 = __hip_atomic_load(at1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else { b0 = *at0; b1 = *at1; }
        return ((b0 | (b1 << 8)) >> (bit & 7)) & 0x1fu;
    };
    // where a value came from (0 G, 1 E, 2 F), by the kind of value: the code's digits, tg the most significant
    const auto cameFrom = [](u32 code, u32 type) -> u32
    {
        const u32 tg = u32(code >= 9) + u32(code >= 18), rest = code - 9 * tg, te = u32(rest >= 3) + u32(rest >= 6), tf = rest - 3 * te;
        return 0 == type ? tg : 1 == type ? te : tf;
    };
    // one row of the band: q = the row's query base, dIn = the database base that enters the band behind it (database[i + 16])
    const auto bandRow = [&](int q, int dIn) -> u32
    {
        const S2 Gv = asS2(G), Ev = asS2(E), Fv = asS2(F);
        // F (:130-173): what cell k hands to cell k + 1 of this row -- max(G - open, E - open, F - extend) and where it came from --
        // computed in place and moved up one cell; cell 0 gets the initial value
        const S2 m = pkMax(Gv, Ev);
        const U2 fE = pkLt(Gv, Ev);
        const S2 v = m - open2, fe = Fv - ext2;
        const int newF = fromCellBelow(asInt(pkMax(v, fe)), initialValue);
        const int tf = fromCellBelow(asInt(pkMaxU(pkLt(v, fe) << 1, fE)), 0);            // v < fe ? 2 : (G < E ? 1 : 0)
        // G (:174-197) with the 16-bit max over byte pairs of the flag vectors: an odd cell takes fF ? 2 : fE, an even cell looks at its
        // odd partner's flags first (the other half of the register)
        const U2 fF = pkLt(m, Fv);
        const S2 gmax = pkMax(m, Fv);
        const U2 twoF = fF << 1;
        const int mx = asInt(pkMaxU(twoF, fE));
        const int negPartnerF = -int(u32(asInt(fF)) >> 16), negPartnerE = -int(u32(asInt(fE)) >> 16);
        const int even = bfi(negPartnerF, asInt(twoF), bfi(negPartnerE, asInt(fE), mx));   // pfF ? 2 fF : (pfE ? fE : max(2 fF, fE))
        const int tg = bfi(0xffff, even, mx);
        // W (:200-244): byte compare, so read 'n' never equals reference 'N'
        const U2 differs = pkNonZero(both(q) ^ d2, ones2);
        const S2 newG = gmax + asS2(differs * wDiff2 + wMatch2);
        // E (:246-297) as an exclusive max-plus suffix scan over the cells: the two cells of a lane in 32 bits, then the lanes.  A lane
        // whose source lies outside the row gets its own value back from the row shift, and max(s, s) = s: only the first step needs to
        // know where the row ends
        const S2 g = newG - open2, f = asS2(newF) - open2;
        const int gf = asInt(pkMax(g, f));
        const int NEG = -(1 << 28);
        const int cLo = s16(gf) - kExtLo, cHi = (gf >> 16) - kExtHi;
        // the maximum over the lanes above: seven v_max whose DPP operand is the same register (a doubling scan is three steps, but each
        // reads through DPP what the step before has just written and waits for it)
        // The first is a DPP move the compiler knows (it keeps the wait states between writing t and reading it through DPP, or fills them);
        // the others come behind it by their dependence on s.
        const int t = max(cLo, cHi);
        int s = fromLaneAbove(t, NEG);
        s = maxWithLaneAbove<4>(s, t); s = maxWithLaneAbove<6>(s, t); s = maxWithLaneAbove<8>(s, t);
        s = maxWithLaneAbove<10>(s, t); s = maxWithLaneAbove<12>(s, t); s = maxWithLaneAbove<14>(s, t);
        const int eLo = max(cHi, s) + kExtHi, eHi = lastLane ? initialValue : s + k1ExtHi;
        const int newE = int((u32(eLo) & 0xffffu) | (u32(eHi) << 16));
        // TE: what cell k + 1 tells cell k -- from its (g, E - ext, f) with the reference's tie rules -- moved down one cell; cell 15 gets 0
        const S2 e1 = asS2(newE) - ext2;
        const U2 fromE = opaque(pkLt(g, e1) & pkLt(f, e1));
        const int teOut = bfi(asInt(-asS2(fromE)) /* per half: all ones where the flag is set */, 0x00010001, asInt(pkLt(g, f) << 1));
        const int te = fromCellAbove(teOut, 0);
        const U2 nine = { 9, 9 }, three = { 3, 3 };
        const U2 code = asU2(tg) * nine + (asU2(te) * three + asU2(tf));       // per half: one of 27
        const int flags = asInt(code);
        G = asInt(newG); E = newE; F = newF;
        // slide the database window: cell k takes cell k - 1's base, cell 0 the next one (what it takes in the last row is not looked at)
        d2 = fromCellBelow(d2, dIn);
        return u32((flags & 0x1f) | ((flags >> 11) & 0x3e0));            // the row's ten bits
    };
#if !defined(ISAAC_TIMING_BSW_NO_DP)      // (timing experiments only: builds with these macros give wrong results)
    if constexpr (BSW_FROM_REGISTERS == SOURCE)
    {   // eight rows per step; the next step's bases are asked for before this one's rows run
        u32 q0, q1, d0, d1;
        query.block(0, q0, q1, d0, d1);
        u32 f[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        u32 b = 0;
        for (; 8 * b + 8 <= L; ++b)
        {
            u32 nq0, nq1, nd0, nd1;
            query.block(b + 1, nq0, nq1, nd0, nd1);
            f[0] = bandRow(int(q0 & 0xffu), int(d0 & 0xffu));
            f[1] = bandRow(int((q0 >> 8) & 0xffu), int((d0 >> 8) & 0xffu));
            f[2] = bandRow(int((q0 >> 16) & 0xffu), int((d0 >> 16) & 0xffu));
            f[3] = bandRow(int(q0 >> 24), int(d0 >> 24));
            f[4] = bandRow(int(q1 & 0xffu), int(d1 & 0xffu));
            f[5] = bandRow(int((q1 >> 8) & 0xffu), int((d1 >> 8) & 0xffu));
            f[6] = bandRow(int((q1 >> 16) & 0xffu), int((d1 >> 16) & 0xffu));
            f[7] = bandRow(int(q1 >> 24), int(d1 >> 24));
            storeFlags(b, f);
            // "used" here, so that the wait for the permutes stands here and counts the stores behind them as allowed to be in flight
            asm volatile("" : "+v"(nq0), "+v"(nq1), "+v"(nd0), "+v"(nd1));
            q0 = nq0; q1 = nq1; d0 = nd0; d1 = nd1;
        }
        if (8 * b < L)
        {   // the last one to seven rows
            u64 qq = u64(q0) | (u64(q1) << 32), dd = u64(d0) | (u64(d1) << 32);
            for (u32 j = 0; 8 * b + j < L; ++j)
            {
                const u32 v = bandRow(int(u32(qq) & 0xffu), int(u32(dd) & 0xffu)); qq >>= 8; dd >>= 8;
#pragma unroll
                for (u32 t = 0; t < 8; ++t) if (t == j) f[t] = v;             // (no private array is indexed at run time)
            }
            storeFlags(b, f);
        }
    }
    else if constexpr (BSW_FROM_LDS == SOURCE)
    {   // staged copies (LDS) with room behind them: four rows' bases per 32-bit read, requested four rows ahead.  (A read per row made
        // every row wait for the row's flag store as well: the waits the compiler places at a loop's head cover everything in flight.)
        const u32 *q4 = reinterpret_cast<const u32 *>(query.q), *d4 = reinterpret_cast<const u32 *>(database + 16);
        u32 qw = q4[0], dw = d4[0], i = 0;
        u32 f[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (; i + 4 <= L; i += 4)
        {
            u32 qn = q4[(i >> 2) + 1], dn = d4[(i >> 2) + 1];
            const u32 h = i & 4;
            if (h)
            {
                f[4] = bandRow(int(qw & 0xffu), int(dw & 0xffu));
                f[5] = bandRow(int((qw >> 8) & 0xffu), int((dw >> 8) & 0xffu));
                f[6] = bandRow(int((qw >> 16) & 0xffu), int((dw >> 16) & 0xffu));
                f[7] = bandRow(int(qw >> 24), int(dw >> 24));
                storeFlags(i >> 3, f);
            }
            else
            {
                f[0] = bandRow(int(qw & 0xffu), int(dw & 0xffu));
                f[1] = bandRow(int((qw >> 8) & 0xffu), int((dw >> 8) & 0xffu));
                f[2] = bandRow(int((qw >> 16) & 0xffu), int((dw >> 16) & 0xffu));
                f[3] = bandRow(int(qw >> 24), int(dw >> 24));
            }
            // "used" here, so that the wait for the two reads stands here and counts the stores behind them as allowed to be in
            // flight; at the loop's head it would wait for everything
            asm volatile("" : "+v"(qn), "+v"(dn));
            qw = qn; dw = dn;
        }
        if (i < L || (i & 4))
        {   // the last one to three rows, and a first half of a block that is still to be stored
            u32 j = i & 7;
            for (; i < L; ++i, ++j)
            {
                const u32 v = bandRow(int(qw & 0xffu), int(dw & 0xffu)); qw >>= 8; dw >>= 8;
#pragma unroll
                for (u32 t = 0; t < 8; ++t) if (t == j) f[t] = v;             // (no private array is indexed at run time)
            }
            storeFlags((i - 1) >> 3, f);
        }
    }
    else
    {
        // the next row's query base and the database base that enters the band with it are requested a row ahead (every lane reads the
        // same bytes: one broadcast access, no branch), so that their latency lies behind the row's arithmetic
        int qNext = u8(query(0)), dNext = u8(database[L > 1 ? 16 : 15]);
        for (u32 i = 0; i < L; i += 8)
        {
            u32 f[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            for (u32 j = 0; j < 8 && i + j < L; ++j)
            {
                const int q = qNext, dIn = dNext;
                const u32 ahead = i + j + 1 < L ? i + j + 1 : i + j;      // the values fetched in the last row are not used
                qNext = u8(query(ahead));
                dNext = u8(database[ahead + 1 < L ? ahead + 16 : ahead + 15]);
                f[j] = bandRow(q, dIn);
            }
            storeFlags(i >> 3, f);
        }
    }
#endif
    reinterpret_cast<int *>(endVals)[l] = G; reinterpret_cast<int *>(endVals)[8 + l] = E; reinterpret_cast<int *>(endVals)[16 + l] = F;
    between();
    STAMP(55);
    if constexpr (GLOBAL_FLAGS) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u32 ret = 0;
#if !defined(ISAAC_TIMING_BSW_NO_TRACEBACK)
    {
        // end-cell scan (:349-379), traceback (:381-435), stripping of the terminal deletions (:437-453).  All 8 lanes of the group
        // walk the traceback together (the same values in every lane; lane 0 does the stores): a stretch of ALIGN cells whose flags
        // say "came from ALIGN" -- nearly all of a typical alignment -- is crossed 16 rows at a time, every lane looking at two rows,
        // instead of one dependent LDS read per row.
        const u32 first = n;
        int mx = s16(int(u16(endVals[15])) - 1);
        int ii = int(L) - 1, jj = ii; u32 maxType = 0;
        for (int cell = 15; cell >= 0; --cell)
            for (u32 type = 0; type < 3; ++type)
            {
                const int value = endVals[16 * type + cell];
                if (value > mx) { mx = value; jj = cell; maxType = type; }
            }
        u32 opLength = 0, firstOp = 0, lastOp = 0;
        const u32 groupShift = ((threadIdx.x & 63u) & ~15u) | (threadIdx.x & 1u);      // the group's lanes: every second bit from here
#define ISAAC_BSW_PUSH(len, op) do { if (n < cap) { lastOp = cigarOp(u32(len), op); if (n == first) firstOp = lastOp; if (l == 0) cig[n] = lastOp; ++n; } else overflow = true; } while (0)
        if (jj > 0) ISAAC_BSW_PUSH(jj, OP_DELETE);
        while (ii >= 0 && jj >= 0 && jj <= 15)
        {
            if (0 == maxType)
            {
                // lane l looks at rows ii - 2l and ii - 2l - 1: bit t of `mask` says that row ii - t ends the run
                const int row = ii - 2 * int(l);
                const bool stop0 = row < 0 || flagsAt(row, jj) >= 9, stop1 = row < 1 || flagsAt(row - 1, jj) >= 9;      // tg != 0
                const u32 mask = (u32(__ballot(stop0) >> groupShift) & 0x5555u) | ((u32(__ballot(stop1) >> groupShift) & 0x5555u) << 1);
                const u32 run = mask ? u32(__ffs(int(mask))) - 1 : 16u;
                opLength += run; ii -= int(run);
                if (!mask) continue;
                if (ii < 0) break;
            }
            ++opLength;
            const u32 nextMaxType = cameFrom(flagsAt(ii, jj), maxType);
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
        if (l == 0) for (u32 lo = first, hi = n; lo + 1 < hi; ++lo) { --hi; const u32 tt = cig[lo]; cig[lo] = cig[hi]; cig[hi] = tt; }
        if (n > first && OP_DELETE == cigarCode(firstOp)) --n;
    }
#endif
    STAMP(56);
    // the group's LDS is reused by the next problem only after lane 0 is done with it
    __builtin_amdgcn_wave_barrier();
    return ret;
}

struct PlainQuery { const char *q; __device__ char operator()(u32 i) const { return q[i]; } };

__global__ __launch_bounds__(BSW_BLOCK) void k_bsw_batch(int matchScore, int mismatchScore, int gapOpenScore, int gapExtendScore,
                                                  const char *sequences, const isaac_bsw_job *jobs, u32 nJobs, u32 maxQueryLength,
                                                  isaac_bsw_result *results)
{
    extern __shared__ __align__(16) u8 lds[];
    const u32 group = bswGroupOfThread(), k = bswLaneOfThread();
    const u32 job = blockIdx.x * (blockDim.x / BSW_GROUP_LANES) + group;
    if (job >= nJobs) return;
    u8 *T = lds + group * bswGroupLdsBytes(maxQueryLength);
    short *endVals = reinterpret_cast<short *>(T + bswFlagBytes(maxQueryLength));
    const isaac_bsw_job jb = jobs[job];
    PlainQuery q; q.q = sequences + jb.query_offset;
    isaac_bsw_result &res = results[job];
    u32 n = 0; bool overflow = false;
    const u32 ret = bswCooperative<BSW_FROM_FUNCTION>(matchScore, mismatchScore, gapOpenScore, gapExtendScore, q, jb.query_length, sequences + jb.database_offset, T, endVals, k,
                                   res.cigar, ISAAC_GPU_MAX_CIGAR_OPS, n, overflow);
    if (k == 0) { res.n_ops = overflow ? 0xffffffffu : n; res.offset = ret; }
}

struct StrandQueryDev { ReadView read; bool reverse; u32 offset; __device__ char operator()(u32 i) const { return strandBase(read, reverse, offset + i); } };

// GappedAligner::alignGapped (GappedAligner.cpp:167-249) for a list of candidates, 8 lanes per candidate: the statements of
// alignGapped() in aligner.h with the DP on the group and everything else on its lane 0.  `bcl` is the tile, the job's
// cluster index is relative to clusterBase.  Grid-stride over the jobs, so the launch does not need the job count on the host.
// NCH: 64-bit registers per lane for either sequence (eight bases each, eight lanes: sequences of up to 64 NCH bases stay in registers and the LDS of a
// group is its traceback flags and end values alone); 0: the sequences are staged in LDS (reads of any length).
template <u32 NCH>
__device__ inline void gappedJobsBody(const DevParams &P, const DevReference &R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                      u32 maxReadLength, GappedResult *results, u8 *flagsArena, u8 *lds)
{
    const bool STAGED = 0 == NCH;
    const u32 FETCH = STAGED ? 4 : NCH;          // chunks per lane a fetch brings
    const u32 group = bswGroupOfThread(), k = bswLaneOfThread(), groups = blockDim.x / BSW_GROUP_LANES;
    // the group's LDS: end values, [the staged sequences,] the traceback flags of its problem (ISAAC_BSW_GLOBAL_FLAGS: those in a region of the arena per group of the grid)
    u8 *mine = lds + group * gappedGroupLdsBytes(maxReadLength, STAGED);
    const u32 stagedBytes = STAGED ? 2 * ((maxReadLength + 47) & ~15u) : 0u;
    u8 *T = ISAAC_BSW_GLOBAL_FLAGS ? flagsArena + (size_t(blockIdx.x) * groups + group) * bswFlagBytes(maxReadLength) : mine + 128 + stagedBytes;
    short *endVals = reinterpret_cast<short *>(mine);
    char *stagedQuery = reinterpret_cast<char *>(mine + 128);
    char *stagedDatabase = stagedQuery + ((maxReadLength + 47) & ~15u);
    const u32 nJobs = imin(*jobCounter, jobsCap);
    // A problem is set up in two steps that each wait for memory -- its record, then the read's bytes and the window of the contig its record points to --
    // and a group that does them in front of its rows stands still for both (12 % of the kernel with two wavefronts per SIMD to hide it behind).  So the
    // group's next problem is fetched beside the current one: its record while the rows run, its bases (into registers) while the group walks the
    // traceback; what is left in front of the rows is the conversion of the read's bytes (and, staged, the stores into LDS).  The loads are volatile so that
    // they stay where they are written: the compiler otherwise moves a load to the block that uses it, behind the loop it was meant to overlap.
    struct Prepared
    {
        Cand f; ReadView read; i64 begin, end, strandPosition; u32 sequenceLength, left, right, cluster; const char *database; bool go;
        u32 queryChunks, window, windowChunks;
    };
    typedef u64 __attribute__((aligned(1))) UnalignedU64;
    const auto fetchJob = [&](u32 j, uint4 (&raw)[5])
    {
        const volatile uint4 *from = reinterpret_cast<const volatile uint4 *>(jobs + j);
#pragma unroll
        for (u32 i = 0; i < 5; ++i) { raw[i].x = from[i].x; raw[i].y = from[i].y; raw[i].z = from[i].z; raw[i].w = from[i].w; }
    };
    static_assert(sizeof(GappedJob) == 80, "five 16-byte pieces");
    const auto prepare = [&](const uint4 (&raw)[5], Prepared &p)
    {
        GappedJob jb; memcpy(&jb, raw, sizeof(jb));
        p.f = jb.in; p.cluster = jb.cluster;
        const u32 r = p.f.readIndex;
        p.read.bcl = bcl + u64(clusterBase + jb.cluster) * P.clusterLength + P.readOffset[r]; p.read.length = P.readLength[r];
        p.read.firstCycle = P.firstCycle[r]; p.read.endCyclesMasked = jb.endCyclesMasked;
        CigarPool none; none.words = nullptr; none.used = 0; none.capacity = 40; none.overflow = 0;       // (the problem's candidate comes without a CIGAR: nothing is read through it)
        candResetAlignment(p.f, none);
        p.f.lowClipped = 0; p.f.highClipped = 0;
        const u64 referenceSize = contigLength(R, p.f.contigId);
        p.go = clipSequence(p.read, p.f, i64(referenceSize), p.begin, p.end, &R, jb.adapterRange);
        p.sequenceLength = 0; p.strandPosition = 0; p.left = 0; p.right = 0; p.database = nullptr; p.queryChunks = 0; p.window = 0; p.windowChunks = 0;
        if (p.go)
        {
            p.sequenceLength = u32(p.end - p.begin);
            p.strandPosition = p.f.position;
            if (i64(referenceSize) < i64(p.sequenceLength) + p.strandPosition + i64(BSW_WIDEST_GAP_SIZE)) p.go = false;
            if (!p.sequenceLength || p.sequenceLength > maxReadLength) p.go = false;
        }
        if (p.go)
        {
            getFlanks(p.strandPosition, p.sequenceLength, referenceSize, p.left, p.right);
            p.database = R.bases + R.contigOffset[p.f.contigId] + p.strandPosition - p.left;
            p.queryChunks = (p.sequenceLength + 7) / 8; p.window = p.sequenceLength + 15; p.windowChunks = (p.window + 7) / 8;
        }
    };
    // the group's 8 lanes bring the query and the window in side by side, eight bases per lane and step, all of a lane's loads requested before
    // the first is used: chunks c0 + k + 8 t (t < FETCH) of either sequence
    const auto fetchChunks = [&](const Prepared &p, u32 c0, u64 (&q)[FETCH], u64 (&w)[FETCH])
    {
        const bool reverse = p.f.reverse != 0;
#pragma unroll
        for (u32 t = 0; t < FETCH; ++t)
        {
            u32 c = c0 + k + BSW_GROUP_LANES * t;
            {   // BCL bytes of strand positions begin + 8c .. + 7 (position t of the chunk in byte t); never reads outside the read's bytes
                const u32 cq = c < p.queryChunks ? c : p.queryChunks - 1;
                const u32 s0 = u32(p.begin) + 8 * cq;
                u64 bytes;
                if (!reverse) { const u32 at = s0 + 8 <= p.read.length ? s0 : p.read.length - 8; bytes = *reinterpret_cast<const volatile UnalignedU64 *>(p.read.bcl + at); bytes >>= 8 * (s0 - at); }
                else
                {   // strand position p is byte length - 1 - p
                    const i32 lo = i32(p.read.length) - 8 - i32(s0);
                    bytes = *reinterpret_cast<const volatile UnalignedU64 *>(p.read.bcl + (lo < 0 ? 0 : lo));
                    if (lo < 0) bytes <<= 8 * u32(-lo);
                    bytes = __builtin_bswap64(bytes);
                }
                q[t] = bytes;
            }
            {
                const u32 cw = c < p.windowChunks ? c : p.windowChunks - 1;
                const u32 at = 8 * cw + 8 <= p.window ? 8 * cw : p.window - 8;       // the window is at least 15 bases
                w[t] = *reinterpret_cast<const volatile UnalignedU64 *>(p.database + at) >> (8 * (8 * cw - at));
            }
        }
    };
    // strandBase for the eight positions of a chunk at once: BCL bytes -> ASCII ACGTn
    const auto asciiChunk = [](u64 bclBytes, bool reverse) -> u64
    {
        const u64 nFlags = zeroBytes(bclBytes & (0xfc * BYTES_01));
        u64 codes = bclBytes & (0x03 * BYTES_01);
        if (reverse) codes ^= 0x03 * BYTES_01;
        const u64 nBytes = (nFlags >> 7) * 0xff;
        return (asciiOfCodes(codes) & ~nBytes) | ((0x6e * BYTES_01) & nBytes);
    };
    const auto stageChunks = [&](const Prepared &p, u32 c0, const u64 (&q)[FETCH], const u64 (&w)[FETCH])
    {
        const bool reverse = p.f.reverse != 0;
#pragma unroll
        for (u32 t = 0; t < FETCH; ++t)
        {
            const u32 c = c0 + k + BSW_GROUP_LANES * t;
            if (c < p.queryChunks) reinterpret_cast<u64 *>(stagedQuery)[c] = asciiChunk(q[t], reverse);
            if (c < p.windowChunks) reinterpret_cast<u64 *>(stagedDatabase)[c] = w[t];
        }
    };
    const u32 stride = gridDim.x * groups;
    u32 j = blockIdx.x * groups + group;
    Prepared cur; cur.go = false;
    u64 q[FETCH], w[FETCH];
#pragma unroll
    for (u32 t = 0; t < FETCH; ++t) { q[t] = 0; w[t] = 0; }
    if (j < nJobs)
    {
        uint4 raw[5]; fetchJob(j, raw);
        prepare(raw, cur);
        if (cur.go) fetchChunks(cur, 0, q, w);
    }
    for (; j < nJobs; j += stride)
    {
        STAMP_BEGIN();
        const u32 jn = j + stride;
        const bool haveNext = jn < nJobs;
        uint4 rawNext[5];
        if (haveNext) fetchJob(jn, rawNext);                       // on its way while this problem's rows run
        Prepared next; next.go = false;
        u64 qn[FETCH], wn[FETCH];
#pragma unroll
        for (u32 t = 0; t < FETCH; ++t) { qn[t] = 0; wn[t] = 0; }
        const auto lookAhead = [&]() { if (haveNext) { prepare(rawNext, next); if (next.go) fetchChunks(next, 0, qn, wn); } };
        GappedResult &res = results[j];
        Cand f = cur.f;
        u32 n = 0, matchCount = 0; bool overflow = false;
        if (cur.go)
        {
            if (cur.begin) { if (k == 0) res.cigar[0] = cigarOp(u32(cur.begin), OP_SOFT_CLIP); n = 1; }
            STAMP(50);
            u32 ret;
            if constexpr (STAGED)
            {
                stageChunks(cur, 0, q, w);
                for (u32 c0 = 4 * BSW_GROUP_LANES; c0 < cur.windowChunks; c0 += 4 * BSW_GROUP_LANES)
                {   // (reads beyond 241 bases: the rest of the sequences is fetched here)
                    u64 qm[FETCH], wm[FETCH];
                    fetchChunks(cur, c0, qm, wm);
                    stageChunks(cur, c0, qm, wm);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                STAMP(51);
                PlainQuery pq; pq.q = stagedQuery;
                ret = bswCooperative<BSW_FROM_LDS, 0 != ISAAC_BSW_GLOBAL_FLAGS>(P.gapMatch, P.gapMismatch, -P.gapOpen, -P.gapExtend, pq, cur.sequenceLength, stagedDatabase, T, endVals, k, res.cigar, 40u, n, overflow,
                                                                                  lookAhead);
            }
            else
            {
                BswRegisterWords<NCH> words;
                const bool reverse = cur.f.reverse != 0;
#pragma unroll
                for (u32 t = 0; t < NCH; ++t) { words.q[t] = asciiChunk(q[t], reverse); words.w[t] = w[t]; }
                words.addressBase = ((threadIdx.x & 63u) & ~14u) << 2;
                STAMP(51);
                ret = bswCooperative<BSW_FROM_REGISTERS, 0 != ISAAC_BSW_GLOBAL_FLAGS>(P.gapMatch, P.gapMismatch, -P.gapOpen, -P.gapExtend, words, cur.sequenceLength, nullptr, T, endVals, k, res.cigar, 40u, n, overflow,
                                                                                        lookAhead);
            }
            STAMP(52);
            if (k == 0)
            {
                i64 strandPosition = cur.strandPosition + ret;
                const u32 clipEndBases = u32(i64(cur.read.length) - cur.end);
                if (clipEndBases) { if (n < 40) res.cigar[n++] = cigarOp(clipEndBases, OP_SOFT_CLIP); else overflow = true; }
                strandPosition -= cur.left;
                // the rescan of the CIGAR (a serial fp64 chain per alignment) is k_gapped_rescan's, one thread per problem:
                // here it would occupy one lane in sixteen.  Handed over: the strand position and "aligned" in matchCount.
                f.position = strandPosition;
                matchCount = 1;
            }
        }
        else lookAhead();
        STAMP(53);
        if (k == 0) { res.out = f; res.matchCount = matchCount; res.nCigar = overflow ? 0xffffffffu : n; }
        STAMP(54);
        cur = next;
#pragma unroll
        for (u32 t = 0; t < FETCH; ++t) { q[t] = qn[t]; w[t] = wn[t]; }
    }
}
// reads of up to 177 bases (2 x 150 and below): three registers per sequence and lane
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                    u32 maxReadLength, GappedResult *results, u8 *flagsArena)
{
    extern __shared__ __align__(16) u8 lds[];
    gappedJobsBody<3>(P, R, bcl, clusterBase, jobs, jobCounter, jobsCap, maxReadLength, results, flagsArena, lds);
}
// ... of up to 305 bases (2 x 250): five
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs_long(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                         u32 maxReadLength, GappedResult *results, u8 *flagsArena)
{
    extern __shared__ __align__(16) u8 lds[];
    gappedJobsBody<5>(P, R, bcl, clusterBase, jobs, jobCounter, jobsCap, maxReadLength, results, flagsArena, lds);
}
// ... of any length: the sequences staged in LDS
__global__ __launch_bounds__(BSW_BLOCK) void k_gapped_jobs_staged(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                           u32 maxReadLength, GappedResult *results, u8 *flagsArena)
{
    extern __shared__ __align__(16) u8 lds[];
    gappedJobsBody<0>(P, R, bcl, clusterBase, jobs, jobCounter, jobsCap, maxReadLength, results, flagsArena, lds);
}

// AlignerBase::updateFragmentCigar for the alignments k_gapped_jobs produced: one thread per problem
__global__ __launch_bounds__(256) void k_gapped_rescan(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, const GappedJob *jobs, const u32 *jobCounter, u32 jobsCap,
                                                      GappedResult *results)
{
    __shared__ double qualityTables[128];
    for (u32 qi = threadIdx.x; qi < 128; qi += blockDim.x) qualityTables[qi] = qi < 64 ? Rg.logMatch[qi] : Rg.logMismatch[qi - 64];
    __syncthreads();
    DevReference R = Rg; R.logMatch = qualityTables; R.logMismatch = qualityTables + 64; R.logStride = 1;
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
