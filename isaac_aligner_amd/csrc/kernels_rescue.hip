// kernels_rescue.hip: see kernels.h and DESIGN.md §4
#include "kernels.h"
#include "template_lean.h"

// k_plan_rescue: the mate-rescue problems of every cluster (template_lean.h: leanPlanCluster), one thread per cluster
#ifndef ISAAC_WAVES_PLAN
#define ISAAC_WAVES_PLAN 0
#endif
#if ISAAC_WAVES_PLAN
__attribute__((amdgpu_waves_per_eu(ISAAC_WAVES_PLAN, ISAAC_WAVES_PLAN)))
#endif
__global__ __launch_bounds__(SELECT_BLOCK) void k_plan_rescue(const TemplateConstants *__restrict__ constants, DevReference R, u32 clusterBase, u32 nChunk, ClusterPools pools, RescueBuffers rb, const u32 *__restrict__ order)
{
#if ISAAC_PLAN_STAGE
    __shared__ u64 stage[SELECT_BLOCK * PLAN_STAGE_WORDS];       // the thread's copy of its cluster's candidate lists: see k_select
#endif
    const DevParams &P = constants->P;
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inChunk = slot < nChunk;
    const u32 t = inChunk ? (order ? order[slot] : slot) : 0;        // clusters of a kind next to each other: see k_cluster_kinds
    const ClusterMeta meta = pools.meta[t];
    // Every seeded candidate is an orphan at most once, so their number bounds the cluster's rescue problems: the slots are
    // reserved first -- one bump of the counter per wavefront -- and the template logic runs once, writing the problems as it meets them
    // (unused slots stay invalid)
    const u32 reserve = (inChunk && meta.built) ? u32(meta.nCands[0]) + meta.nCands[1] : 0;
    u32 incl = reserve;
    for (u32 o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += v; }
    const u32 waveTotal = __shfl(incl, 63, 64);
    u32 base = 0, n = 0;
    if ((threadIdx.x & 63) == 63 && waveTotal) base = atomicAdd(rb.jobCounter, waveTotal);
    base = __shfl(base, 63, 64) + incl - reserve;
    if (!inChunk) return;
    if (reserve)
    {
        // what does not fit: the cluster runs its rescues itself in the wave-per-cluster pass
        if (base + reserve > rb.jobsCap)
        {   // (the slots of this reservation that do lie below the capacity belong to nobody: the kernels that walk the slots must not take what an earlier chunk left there for problems)
            for (u32 i = base; i < rb.jobsCap; ++i) { rb.jobActive[i] = 0; rb.jobs[i].valid = 0; }
            base = 0xffffffffu;
        }
        RescueJob *jobs = 0xffffffffu == base ? nullptr : rb.jobs + base;
        LeanCtx x;
        x.P = &P; x.R = &R; x.tls = &constants->tls;
        x.l0 = pools.cands + meta.first; x.l1 = x.l0 + meta.second; x.n0 = meta.nCands[0]; x.n1 = meta.nCands[1];
        x.pool = pools.cigars + 3 * u64(meta.first);
        x.rogRead0 = 0.0; x.rogRead1 = 0.0; x.rog = 0.0; x.logMismatchQ40 = 0.0;      // no score is computed here
        x.clusterId = clusterBase + t; x.mapqNearInteger = 0;
#if ISAAC_PLAN_STAGE
        if (reserve <= PLAN_STAGE)
        {
            Cand *mine = reinterpret_cast<Cand *>(stage + threadIdx.x * PLAN_STAGE_WORDS);
            uint4 v[PLAN_STAGE][4];
#pragma unroll
            for (u32 i = 0; i < PLAN_STAGE; ++i)
                if (i < reserve)
                {
                    const uint4 *from = reinterpret_cast<const uint4 *>(i < x.n0 ? x.l0 + i : x.l1 + (i - x.n0));
                    v[i][0] = from[0]; v[i][1] = from[1]; v[i][2] = from[2]; v[i][3] = from[3];
                }
#pragma unroll
            for (u32 i = 0; i < PLAN_STAGE; ++i)
                if (i < reserve)
                {
                    uint2 *to = reinterpret_cast<uint2 *>(mine + i);
                    to[0] = make_uint2(v[i][0].x, v[i][0].y); to[1] = make_uint2(v[i][0].z, v[i][0].w); to[2] = make_uint2(v[i][1].x, v[i][1].y); to[3] = make_uint2(v[i][1].z, v[i][1].w);
                    to[4] = make_uint2(v[i][2].x, v[i][2].y); to[5] = make_uint2(v[i][2].z, v[i][2].w); to[6] = make_uint2(v[i][3].x, v[i][3].y); to[7] = make_uint2(v[i][3].z, v[i][3].w);
                }
            x.l0 = mine; x.l1 = mine + x.n0;
            n = leanPlanCluster(x, t, jobs);
        }
        else
#endif
        n = leanPlanCluster(x, t, jobs);
        if (jobs)
        {
            for (u32 i = n; i < reserve; ++i) { jobs[i].valid = 0; jobs[i].fallback = 0; jobs[i].nCands = 0; jobs[i].nGapped = 0; }
            for (u32 i = 0; i < n; ++i)
            {
                if (!jobs[i].valid) continue;
                const u32 words = (jobs[i].windowLen + P.readLength[jobs[i].shadowReadIndex] + 31) / 32;
                if (words <= RW_LDS_BITMAP) continue;               // k_rescue_windows keeps short bitmaps in LDS
                const u32 at = atomicAdd(rb.bitmapCounter, words);
                if (at + words > rb.bitmapCap) jobs[i].fallback = 1; else { jobs[i].bitmapBase = at; jobs[i].bitmapWords = words; }
            }
            for (u32 i = 0; i < reserve; ++i) rb.jobActive[base + i] = i < n && jobs[i].valid && !jobs[i].fallback;
        }
    }
    rb.jobBase[t] = base; rb.jobCount[t] = n;
}


// The bases of window positions [g, g + 32) for one lane: their 2-bit codes (position g in the low bits) and their not-ACGT flags, from
// the packed copy of the reference.  g is an index into the concatenated contigs.  (Three words and two funnel shifts: the third
// word is not needed when g is a multiple of 16, but a branch around one load costs more than the load.)
struct WindowBits { u64 codes; u32 notBase; };
__device__ inline WindowBits loadWindowBits(const DevReference &R, u64 g)
{
    WindowBits w;
    g = g < R.totalBases ? g : R.totalBases;          // lanes past the end of a window that ends the reference: read the spare words
    const u32 *pw = R.packedBases + (g >> 4);
    const u32 w0 = pw[0], w1 = pw[1], w2 = pw[2];
    const u32 shift = 2 * u32(g & 15);
    w.codes = u64(__builtin_amdgcn_alignbit(w1, w0, shift)) | (u64(__builtin_amdgcn_alignbit(w2, w1, shift)) << 32);
    const u32 *pn = R.notBase + (g >> 5);
    w.notBase = __builtin_amdgcn_alignbit(pn[1], pn[0], u32(g & 31));
    return w;
}
static_assert(RW_PER_LANE + 6 <= 32, "a lane's window bases fit the words loadWindowBits returns");
// The same for window position `offset` of a window whose first base has index `windowBase` in the concatenated contigs, the same in every lane: the 64-bit part
// of the address arithmetic is the wave's (scalar registers), a lane adds a 32-bit word offset; the clamp at the reference's end is a minimum with what is left
// of the reference behind the window's first base.  (No branch: two paths that meet again would each have to finish their loads where they stand.)
__device__ inline WindowBits loadWindowBits(const DevReference &R, u64 windowBase, u32 offset)
{
    WindowBits w;
    const u64 room = R.totalBases - windowBase;                       // (windows begin inside the reference)
    offset = imin(offset, room > 0xffffffffull ? 0xffffffffu : u32(room));
    const u32 o = u32(windowBase & 15u) + offset;
    const u32 *pw = R.packedBases + (windowBase >> 4);
    const u32 at = o >> 4;
    const u32 w0 = pw[at], w1 = pw[at + 1], w2 = pw[at + 2];
    const u32 shift = 2 * (o & 15u);
    w.codes = u64(__builtin_amdgcn_alignbit(w1, w0, shift)) | (u64(__builtin_amdgcn_alignbit(w2, w1, shift)) << 32);
    const u32 on = u32(windowBase & 31u) + offset;
    const u32 *pn = R.notBase + (windowBase >> 5);
    const u32 atn = on >> 5;
    w.notBase = __builtin_amdgcn_alignbit(pn[atn + 1], pn[atn], on);
    return w;
}

// k_rescue_windows: one wave per rescue problem (ShadowAligner::findShadowCandidatePositions, ShadowAligner.cpp:53-112).
// The mate's 7-mers go to an LDS hash table (first read position per k-mer).  The window is walked in tiles of 64 x RW_PER_LANE bases:
// every lane takes RW_PER_LANE consecutive positions; with the reference 2 bits per base a 7-mer is a shift and a mask of the
// lane's word (and seven zero bits of the not-ACGT map), and each one is looked up.  "Push unless equal to the previous hit's candidate" needs the previous hit in window order: inside a lane that is
// sequential, across lanes one ballot + shuffle, across tiles a carried value.  Pushed candidates set bits in a bitmap
// (LDS for ordinary windows, global for the long ones), whose ascending enumeration is the reference's sort + unique.
template <bool LDS_BITMAP>
__device__ inline void rescueWindowScan(const DevReference &R, const RescueJob &job, u64 windowBase, const WindowBits &firstTile, u32 L, const u32 *tab, u32 *bitmap, u32 lane, u32 &pushes)
{
    const i32 bias = i32(L) - 7;
    const i32 lastStart = i32(job.windowLen) - 7;       // last valid k-mer start
    i32 carry = 0; bool haveCarry = false;
    for (i32 tile = 0; tile * RW_TILE <= lastStart; ++tile)
    {
        const i32 p0 = tile * RW_TILE + i32(lane) * i32(RW_PER_LANE);          // window position of this lane's first base
        const WindowBits wb = tile ? loadWindowBits(R, windowBase + u64(p0)) : firstTile;
        i32 cand[RW_PER_LANE]; u32 hitMask = 0;
#pragma unroll
        for (u32 k = 0; k < RW_PER_LANE; ++k)
        {
            const i32 p = p0 + i32(k);
            cand[k] = 0;
            if (p <= lastStart && !((wb.notBase >> k) & 0x7fu))
            {
                const u32 kmer = u32(wb.codes >> (2 * k)) & 0x3fffu;
                u32 h = (kmer * 2654435761u) >> 23;
                while (true)
                {
                    const u32 e = tab[h];
                    if (e == KMER_EMPTY) break;
                    if ((e >> 10) == kmer) { hitMask |= 1u << k; cand[k] = p - i32(e & 0x3ffu); break; }
                    h = (h + 1) & (RW_TABLE - 1);
                }
            }
        }
        // previous hit in window order for this lane's first hit
        i32 lastCand = 0;
#pragma unroll
        for (u32 k = 0; k < RW_PER_LANE; ++k) if (hitMask & (1u << k)) lastCand = cand[k];
        const unsigned long long lanesWithHits = __ballot(hitMask != 0);
        const unsigned long long below = lanesWithHits & ((1ull << lane) - 1ull);
        const int prevLane = below ? 63 - __clzll(below) : 0;
        const i32 prevCand = __shfl(lastCand, prevLane, 64);
        bool havePrev = below ? true : haveCarry;
        i32 prev = below ? prevCand : carry;
        u32 localPushes = 0;
#pragma unroll
        for (u32 k = 0; k < RW_PER_LANE; ++k)
            if (hitMask & (1u << k))
            {
                if (!havePrev || prev != cand[k])
                {
                    ++localPushes;
                    const u32 bit = u32(cand[k] + bias);
                    atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
                }
                prev = cand[k]; havePrev = true;
            }
        for (int o = 32; o > 0; o >>= 1) localPushes += __shfl_xor(localPushes, o, 64);
        pushes += localPushes;
        if (lanesWithHits) { carry = __shfl(lastCand, 63 - __clzll(lanesWithHits), 64); haveCarry = true; }
    }
}

// The ordinary window (candidate bitmap in LDS, far fewer than SHADOW_POSITIONS_MAX positions): which 7-mers the mate has at all is a
// 16384-bit map, so a window position costs one LDS word and a bit test, and only the positions whose 7-mer does occur in the mate (about
// 1 % of a window outside the mate's own place) go on to the hash table for the read offset.
// (Round 3 tried a map addressed by the six bases two neighbouring positions share, a byte answering both -- half the LDS reads, which hit
// random banks -- and lost: 4.9 -> 6.0 ms per 1 M clusters.  The map doubles to 4 KB per wave, fewer workgroups fit a CU, and the kernel
// lives on its occupancy: the bank conflicts it removes were hidden behind other waves' arithmetic.)  The probe loops of the general form run
// once per position and lane *for the whole wave*; here they run for the few hits.  The reference's "push unless equal to the previous
// candidate" only decides how many entries its position list holds before sort + unique; that number cannot reach the list's capacity in
// a window this short, and the set of candidates is the same without it.
typedef const __attribute__((address_space(3))) u32 LdsWord;
static const u32 RW_DENSE_HITS = 6;
// the lane that holds tile position q when every lane holds PL consecutive ones (q < 64 * PL <= 1024)
template <u32 PL> __device__ inline u32 tileLaneOf(u32 q)
{
    static_assert(PL == 8 || PL == 12 || PL == 16, "tile sizes of rescueWindowScanShort");
    return PL == 8 ? q >> 3 : PL == 16 ? q >> 4 : __umul24(q, 43691u) >> 19;       // q / 12 for q < 2^16
}
// The mate's 7-mers of an ordinary window, without a hash table: `present` has one bit per possible 7-mer; a k-mer's rank among the set bits (the set bits in
// front of its 64-bit block, 256 16-bit counts, plus those below it in the block) is its slot in `firstPosition`, which holds the mate's first position with it
// (ShadowAligner::hashShadowKmers keeps the first, ShadowAligner.cpp:53-72).  Building it costs no compare-and-swap loops -- the hash table's probe sequences ran
// for the whole wave as long as its unluckiest lane's -- and a hit reads two words and then one, with no loop either.
__device__ inline u32 rescueKmerRank(u32 kmer, const u32 *present, const u16 *blockPrefix)
{
    const u32 block = kmer >> 6;
    const u64 pair = reinterpret_cast<const u64 *>(present)[block];
    return u32(blockPrefix[block]) + u32(__popcll(pair & ((1ull << (kmer & 63u)) - 1ull)));
}
// one window position whose 7-mer the mate has: the mate's first position with it, and the candidate that puts it there
__device__ inline void rescueWindowHit(u32 kmer, i32 biasedPosition, const u32 *firstPosition, const u32 *present, const u16 *blockPrefix, u32 *bitmap)
{
    const u32 bit = u32(biasedPosition - i32(firstPosition[rescueKmerRank(kmer, present, blockPrefix)]));
    atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
}
// inclusive prefix sum over the wave's 64 lanes (all of them active): four shifts inside the rows of 16, then the rows' totals handed on
__device__ inline u32 waveInclusiveAdd(u32 v)
{
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));      // row_shr:1
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));      // row_shr:2
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));      // row_shr:4
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));      // row_shr:8
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
    v += u32(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
    return v;
}
template <u32 PL>
__device__ inline void rescueWindowScanShort(const DevReference &R, const RescueJob &job, u64 windowBase, const WindowBits &firstTile, u32 L, const u32 *firstPosition, const u32 *present,
                                             const u16 *blockPrefix, const u32 *mateCodes, const u32 *firstFlags, u32 *bitmap, u32 lane STAMP_PARAM)
{
    static_assert(PL <= 16, "a lane's bit offsets 2k stay below 32");
    const i32 bias = i32(L) - 7;
    const i32 lastStart = i32(job.windowLen) - 7;       // last valid k-mer start
    u32 presentBase = u32(size_t((LdsWord *)present));                     // a multiple of 2048: OR instead of ADD
    asm("" : "+v"(presentBase));                                           // in a vector register: (x & mask) | base is then one v_and_or_b32 (its mask may be the one scalar operand)
    for (i32 tile = 0; tile * i32(64 * PL) <= lastStart; ++tile)
    {
        const i32 p0 = tile * i32(64 * PL) + i32(lane * PL);               // window position of this lane's first base
        const WindowBits wb = tile ? loadWindowBits(R, windowBase, u32(p0)) : firstTile;
        const u32 lo = u32(wb.codes), hi = u32(wb.codes >> 32);
        // five vector instructions per position: the word's LDS address (bits 2k+5 .. 2k+13 of the codes, times four, into the map's
        // base), the bit's number (the shifter takes the low five bits of bits 2k ..), the shift, and the bit into the mask
        u32 words[PL];
#pragma unroll
        for (u32 k = 0; k < PL; ++k)
        {
            const u32 s = 2 * k + 3;
            const u32 x = s + 11 <= 32 ? lo >> s : s < 32 ? __builtin_amdgcn_alignbit(hi, lo, s) : hi >> (s - 32);
            words[k] = *(LdsWord *)size_t((x & 0x7fcu) | presentBase);
        }
        // positions whose seven bases are all ACGT and which start a k-mer inside the window
        u32 inv = wb.notBase; inv |= inv >> 1; inv |= inv >> 2; inv |= inv >> 3;
        const i32 nValid = imin(imax(lastStart - p0 + 1, 0), i32(PL));
        const u32 valid = ~inv & ((1u << nValid) - 1u);
        u32 h = 0;
#pragma unroll
        for (u32 k = 0; k < PL; ++k)
        {
            const u32 t = k == 0 ? lo : 2 * k + 5 <= 32 ? lo >> (2 * k) : __builtin_amdgcn_alignbit(hi, lo, 2 * k);
            h = __builtin_amdgcn_alignbit(words[k] >> (t & 31u), h, 1);    // bit k ends at 32 - PL + k
        }
        u32 hitMask = (h >> (32 - PL)) & valid;
        STAMP(8);
        // Where the mate really lies every position hits, a dozen neighbouring lanes with PL hits each, and all of them name the same candidate: the window
        // repeats the mate there, base for base.  So every lane looks its first hit up, and asks of the others only whether they continue it: the hit d + 1
        // positions on names the same candidate if the window's 7-mer there is the mate's at the first hit's mate position + d + 1 and that position is the
        // first the mate has this 7-mer at (firstFlags) -- a comparison of the lane's window bases with the mate's bases from there, all positions at once.
        // What that explains is dropped; a lane with a hit it does not explain (another diagonal: an indel, a repeat) keeps all its later hits for the loops below.
        if (__ballot(hitMask != 0))
        {
            const bool have = hitMask != 0;
            const u32 k0 = have ? u32(__ffs(hitMask)) - 1 : 0;
            u32 rest = have ? (hitMask >> k0) >> 1 : 0;                    // bit d: position k0 + 1 + d hits
            u32 first = 0;
            if (have)
            {
                first = firstPosition[rescueKmerRank(u32(wb.codes >> (2 * k0)) & 0x3fffu, present, blockPrefix)];
                const u32 bit = u32(p0 + i32(k0) + bias - i32(first));
                atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
            }
            if (__ballot(rest != 0))
            {
                u32 explained = 0;
                if (rest)
                {
                    const u32 a = first + 1;                               // the mate position that lies against window position k0 + 1
                    const u64 windowCodes = (wb.codes >> (2 * k0)) >> 2;
                    const u32 *mw = mateCodes + (a >> 4);
                    const u32 m0 = mw[0], m1 = mw[1], m2 = mw[2], shift = 2 * (a & 15u);
                    const u32 xl = u32(windowCodes) ^ __builtin_amdgcn_alignbit(m1, m0, shift), xh = u32(windowCodes >> 32) ^ __builtin_amdgcn_alignbit(m2, m1, shift);
                    // bit 2j: base j differs; then, bit 2d: one of bases d .. d+6 differs (d <= 14: the low word, which looks at bases up to 20)
                    const u32 dl = (xl | (xl >> 1)) & 0x55555555u, dh = (xh | (xh >> 1)) & 0x55555555u;
                    const u32 pl = dl | __builtin_amdgcn_alignbit(dh, dl, 2), ph = dh | (dh >> 2);                          // bases d, d+1
                    const u32 ql = pl | __builtin_amdgcn_alignbit(ph, pl, 4);                                               // d .. d+3
                    u32 differs = ql | __builtin_amdgcn_alignbit(ph, pl, 8) | __builtin_amdgcn_alignbit(dh, dl, 12);      // + d+4, d+5 + d+6
                    // the even bits gathered: bit d
                    differs = (differs | (differs >> 1)) & 0x33333333u;
                    differs = (differs | (differs >> 2)) & 0x0f0f0f0fu;
                    differs = (differs | (differs >> 4)) & 0x00ff00ffu;
                    differs = (differs | (differs >> 8)) & 0x0000ffffu;
                    const u32 *fw = firstFlags + (a >> 5);
                    explained = ~differs & __builtin_amdgcn_alignbit(fw[1], fw[0], a & 31u);
                }
                if ((rest & ~explained) == 0) rest = 0;
            }
            hitMask = have ? (rest << k0) << 1 : 0;
        }
        // What is left is little as a rule.  Should a lane still have many (its hits one by one make as many passes of the whole wave), the hits are
        // dealt out again, lane l taking tile positions l, l + 64, ...
        if (__ballot(__popc(hitMask) > RW_DENSE_HITS) == 0)
            while (hitMask)
            {
                const u32 k = u32(__ffs(hitMask)) - 1; hitMask &= hitMask - 1;
                rescueWindowHit(u32(wb.codes >> (2 * k)) & 0x3fffu, p0 + i32(k) + bias, firstPosition, present, blockPrefix, bitmap);
            }
        else
        {
            u32 mine = 0;                                                  // bit t: tile position t * 64 + lane hits
#pragma unroll
            for (u32 t = 0; t < PL; ++t)
            {
                const u32 q = t * 64 + lane, l = tileLaneOf<PL>(q), k = q - l * PL;
                mine = __builtin_amdgcn_alignbit(u32(__shfl(hitMask, l, 64)) >> k, mine, 1);
            }
            mine >>= 32 - PL;
            while (__ballot(mine != 0))                                    // every lane stays in the loop: the exchanges read all lanes
            {
                const bool have = mine != 0;
                const u32 t = have ? u32(__ffs(mine)) - 1 : 0; mine &= mine - 1;
                const u32 q = t * 64 + lane, l = tileLaneOf<PL>(q), k = q - l * PL;
                const u64 codes = u64(u32(__shfl(lo, l, 64))) | (u64(u32(__shfl(hi, l, 64))) << 32);
                if (have) rescueWindowHit(u32(codes >> (2 * k)) & 0x3fffu, tile * i32(64 * PL) + i32(q) + bias, firstPosition, present, blockPrefix, bitmap);
            }
        }
        STAMP(9);
    }
}

// (Round 3: taking the problems from a compacted list of the slots in use -- three in eight; a slot is reserved per seeded candidate --
// so that all four waves of a workgroup work left the kernel where it was, 4.80 against 4.80 ms: the empty waves leave at once and
// were not what kept working waves off the CUs.)
// (A resident grid of 16 K wavefronts striding over the problem slots -- two slots in three are empty, they are reserved per seeded
// candidate -- was measured slower, 8.0 against 6.4 ms per 1 M clusters: the windows differ in length and the hardware's own wave
// scheduling balances them better.)
__device__ inline void rescueWindowsProblem(const DevParams &P, const DevReference &R, const u8 *bcl, u32 clusterBase, const RescueBuffers &rb, u32 j, u32 lane,
                                            u32 *tab, u32 *ldsBitmap, u32 *present, u16 *blockPrefix, u32 *mateCodes, u32 *firstFlags)
{
    // Is there a problem in the slot?  Two slots in three are empty (they are reserved per seeded candidate): a byte per slot, four megabytes that stay in L2, and
    // the number of slots in use say so; the empty wavefronts leave on that.  Then the record, 24 words through the scalar cache, awaited once.  (Left to the
    // compiler the byte-sized fields came by vector loads, the two of them that decide whether there is anything to do first and the rest behind the branch, and the
    // counter in between: three memory latencies in a row before the first useful load was issued.)
    typedef u32 Words16 __attribute__((ext_vector_type(16)));
    typedef u32 Words8 __attribute__((ext_vector_type(8)));
    const u32 slot = imin(j, rb.jobsCap - 1);
    u32 activeBytes, slotsInUse;
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(activeBytes), "=&s"(slotsInUse) : "s"(reinterpret_cast<const u32 *>(rb.jobActive) + (slot >> 2)), "s"(rb.jobCounter) : "memory");
    const u32 nJobs = imin(slotsInUse, rb.jobsCap);
    STAMP_BEGIN();
    if (!(j < nJobs && ((activeBytes >> (8 * (slot & 3u))) & 0xffu))) return;
    Words16 head; Words8 tail;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(head), "=&s"(tail) : "s"(rb.jobs + slot) : "memory");
    static_assert(offsetof(RescueJob, windowLen) == 8 && offsetof(RescueJob, cluster) == 12 && offsetof(RescueJob, bitmapBase) == 32 && offsetof(RescueJob, shadowReadIndex) == 40 &&
                  offsetof(RescueJob, fallback) == 43 && offsetof(RescueJob, windowBaseHigh) == 86 && offsetof(RescueJob, windowBaseLow) == 88, "the words picked below");
    RescueJob job;
    job.windowLen = head[2]; job.cluster = head[3]; job.bitmapBase = head[8];
    job.shadowReadIndex = u8(head[10]); job.shadowReverse = u8(head[10] >> 8); job.valid = u8(head[10] >> 16); job.fallback = u8(head[10] >> 24);
    job.windowBaseHigh = u16(tail[5] >> 16); job.windowBaseLow = tail[6];
    const bool active = true;
    STAMP(0);
    u32 pushes = 0, total = 0, bitmapWords = 0, L = 0, smallWord = 0, smallIncl = 0;
    bool small = true;
    u32 *bitmap = ldsBitmap;
    if (active)
    {
        // the lane's bytes of the first window tile are requested now and used after the k-mer table is built: one memory
        // latency instead of two in a row
        const u32 r = job.shadowReadIndex;
        L = P.readLength[r];
        bitmapWords = (job.windowLen + L + 31) / 32;
        small = bitmapWords <= RW_LDS_BITMAP;
        // positions per lane and tile: the windows of a run are nearly all one length (the template length model's), and a tile that
        // covers the window in one go keeps every lane busy (8 per lane left the second tile of a 710-base window 60 % empty)
        const u32 nStarts = job.windowLen > 6 ? job.windowLen - 6 : 0;
        const u32 perLane = __builtin_amdgcn_readfirstlane(!small ? RW_PER_LANE : nStarts <= 64 * 8 ? 8u : nStarts <= 64 * 12 ? 12u : 16u);
        const u64 windowBase = rescueJobWindowBase(job);
        const WindowBits firstTile = loadWindowBits(R, windowBase, lane * perLane);
        // ... and so are the lane's eight bytes of the mate: they arrive while the tables are cleared
        ReadView read; read.bcl = bcl + u64(clusterBase + job.cluster) * P.clusterLength + P.readOffset[r]; read.length = L; read.endCyclesMasked = 0; read.firstCycle = 0;
        const bool reverse = job.shadowReverse != 0;
        // (one load from a selected address; the last lane's shift waits until the bytes are used: nothing here may wait for them)
        // (every lane loads, the ones past the read's end its first bytes: a load under a condition is taken apart, and waited for, where it stands)
        u64 bytes; u32 shortBy;
        {
            const u32 s0 = lane * 8;                                 // first strand position of this lane
            const bool mine = s0 < L;
            shortBy = mine && s0 + 8 > L ? s0 + 8 - L : 0;            // strand positions of this lane that lie past the read's end
            // forward: BCL bytes s0 .. s0+7 (the last lane: L-8 .. L-1, shifted down); reverse: L-8-s0 .. L-1-s0, strand position s0+t is byte 7-t (the last lane: 0 .. 7, shifted up)
            const u32 from = !mine ? 0 : !reverse ? (shortBy ? L - 8 : s0) : (shortBy ? 0 : L - 8 - s0);
            memcpy(&bytes, read.bcl + from, 8);
        }
        // the slots of the mate's k-mers (an ordinary window: a first position per k-mer in use; a long one: the hash table), all ones either way
        const u32 nKmers = L > 6 ? L - 6 : 0;
        for (u32 i = lane; i < (small ? (nKmers + 3) / 4 : RW_TABLE / 4); i += 64) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(KMER_EMPTY, KMER_EMPTY, KMER_EMPTY, KMER_EMPTY);
        if (!small) bitmap = rb.bitmaps + job.bitmapBase;
        if (small) for (u32 i = lane; i < RW_PRESENT_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(present)[i] = make_uint4(0, 0, 0, 0);
        // (the two address spaces apart: through the one pointer the stores and the count's loads below are flat instructions)
        if (small) { if (lane < bitmapWords) ldsBitmap[lane] = 0; }
        else { for (u32 i = lane; i < bitmapWords; i += 64) bitmap[i] = 0; __threadfence(); }
        __builtin_amdgcn_wave_barrier();
        STAMP(1);
        // the mate's 7-mers: first read position of every k-mer (ShadowAligner::hashShadowKmers, :53-72)
        // The mate in the strand's order, packed once: lane l holds strand positions 8l .. 8l+7 as 2-bit codes (the code the packed reference
        // has for the same base: A 0, C 1, G 3, T 2) in bits 0-15 and their N flags in bits 16-23 (a BCL byte without quality bits is an N,
        // Read.cpp:56-69; so is what lies past the end of the read).  A 7-mer is then 14 bits of two neighbouring lanes' words.
        u32 packedMate = 0x00ff0000u;
        if (lane * 8 < L)
        {
            // eight bytes at once: strand position s0 + t in byte t, the complement for the reverse strand, then per byte the code
            // base ^ (base >> 1) and "no quality bits", and the fields of four bytes gathered with shifts
            asm volatile("" : "+v"(bytes));                                   // (not before: the compiler would start taking the bytes apart right behind the load, and wait for it there)
            bytes = !reverse ? bytes >> (8 * shortBy) : bytes << (8 * shortBy);
            if (reverse) bytes = __builtin_bswap64(bytes) ^ 0x0303030303030303ull;
            u32 codes = 0, ns = 0;
#pragma unroll
            for (u32 half = 0; half < 2; ++half)
            {
                const u32 x = u32(bytes >> (32 * half));
                const u32 base = x & 0x03030303u;
                const u32 c = base ^ ((base >> 1) & 0x01010101u);
                u32 y = (c | (c >> 6)) & 0x000f000fu;                      // bytes 0, 1 -> bits 0-3; bytes 2, 3 -> bits 16-19
                y = (y | (y >> 12)) & 0xffu;
                // a byte's quality is 0..63: adding 63 sets bit 6 unless it is 0
                u32 n = ((((x >> 2) & 0x3f3f3f3fu) + 0x3f3f3f3fu) & 0x40404040u) ^ 0x40404040u;
                n >>= 6; n |= n >> 7; n = (n | (n >> 14)) & 0xfu;
                codes |= y << (8 * half); ns |= n << (4 * half);
            }
            packedMate = codes | (ns << 16);
        }
        if (small)
        {   // the mate's codes in one piece, 16 bases a word (lane l: bases 8l .. 8l+7; past the read's end: zeros, and a spare word for reads that end a word)
            reinterpret_cast<u16 *>(mateCodes)[lane] = u16(packedMate);
            if (lane < RW_MATE_WORDS - 32) mateCodes[32 + lane] = 0;
            if (lane < RW_FLAG_WORDS) firstFlags[lane] = 0;
        }
        STAMP(10);
        // the k-mer that starts at the mate's position i (base k of it at bits 2k, as loadWindowBits lays them out), or all ones: past the end, or an N among
        // its bases.  Every lane takes part in the exchanges.
        const auto mateKmer = [&](u32 i) -> u32
        {
            const u32 w0 = __shfl(packedMate, i >> 3, 64), w1 = __shfl(packedMate, ((i >> 3) + 1) & 63, 64);
            const u32 shift = i & 7;
            const u32 kmer = (((w0 & 0xffffu) | (w1 << 16)) >> (2 * shift)) & 0x3fffu;
            const bool ok = i + 7 <= L && 0 == (((((w0 >> 16) & 0xffu) | (((w1 >> 16) & 0xffu) << 8)) >> shift) & 0x7fu);
            return ok ? kmer : 0xffffffffu;
        };
        // -DISAAC_TIMING_RW_NO_TABLE / _NO_SCAN / _NO_ENUM: builds that leave a section out (wrong results) to time the others, scripts/exp_rw_phases.sh
#if !defined(ISAAC_TIMING_RW_NO_TABLE)
        if (small)
        {
            // which k-mers there are; their ranks; the first position of each.  (Read lengths end at 512: eight rounds of 64 positions at most, the
            // k-mers of a round kept in a register between the first pass and the third.)
            u32 held[8];
#pragma unroll
            for (u32 t = 0; t < 8; ++t)
            {
                held[t] = 0xffffffffu;
                if (t * 64 < L)
                {
                    held[t] = mateKmer(t * 64 + lane);
                    if (held[t] != 0xffffffffu) atomicOr(&present[held[t] >> 5], 1u << (held[t] & 31u));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            STAMP(11);
            {   // lane l counts blocks 2l, 2l+1 (words 4l .. 4l+3) and 128+2l, 128+2l+1: two 16-byte reads a lane, next to each other across the wave
                const uint4 a = reinterpret_cast<const uint4 *>(present)[lane], b = reinterpret_cast<const uint4 *>(present)[64 + lane];
                const u32 a0 = u32(__popc(a.x)) + u32(__popc(a.y)), a1 = u32(__popc(a.z)) + u32(__popc(a.w));
                const u32 b0 = u32(__popc(b.x)) + u32(__popc(b.y)), b1 = u32(__popc(b.z)) + u32(__popc(b.w));
                const u32 inclusive = waveInclusiveAdd((a0 + a1) | ((b0 + b1) << 16));      // (at most 506 k-mers: the low half never carries)
                const u32 lowTotal = u32(__builtin_amdgcn_readlane(int(inclusive), 63)) & 0xffffu;
                const u32 beforeA = (inclusive & 0xffffu) - (a0 + a1), beforeB = (inclusive >> 16) - (b0 + b1) + lowTotal;
                reinterpret_cast<u32 *>(blockPrefix)[lane] = beforeA | ((beforeA + a0) << 16);
                reinterpret_cast<u32 *>(blockPrefix)[64 + lane] = beforeB | ((beforeB + b0) << 16);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            STAMP(12);
#pragma unroll
            for (u32 t = 0; t < 8; ++t)
                if (t * 64 < L && held[t] != 0xffffffffu) { held[t] = rescueKmerRank(held[t], present, blockPrefix); atomicMin(&tab[held[t]], t * 64 + lane); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // firstFlags, bit i: the mate's position i is the first with its 7-mer (rescueWindowScanShort: where a run of hits may be taken for granted)
#pragma unroll
            for (u32 t = 0; t < 8; ++t)
                if (t * 64 < L)
                {
                    const unsigned long long firsts = __ballot(held[t] != 0xffffffffu && tab[held[t] != 0xffffffffu ? held[t] : 0] == t * 64 + lane);
                    if (lane < 2) firstFlags[2 * t + lane] = lane ? u32(firsts >> 32) : u32(firsts);
                }
        }
        else
            for (u32 i = lane; i < ((L + 63) & ~63u); i += 64)
            {
                const u32 kmer = mateKmer(i);
                if (kmer == 0xffffffffu) continue;
                const u32 val = (kmer << 10) | i;
                u32 h = (kmer * 2654435761u) >> 23;
                while (true)
                {
                    const u32 old = atomicCAS(&tab[h], KMER_EMPTY, val);
                    if (old == KMER_EMPTY) break;
                    if ((old >> 10) == kmer) { atomicMin(&tab[h], val); break; }
                    h = (h + 1) & (RW_TABLE - 1);
                }
            }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(2);
#if !defined(ISAAC_TIMING_RW_NO_TABLE) && !defined(ISAAC_TIMING_RW_NO_SCAN)
        if (small)
        {
            if (perLane == 8) rescueWindowScanShort<8>(R, job, windowBase, firstTile, L, tab, present, blockPrefix, mateCodes, firstFlags, ldsBitmap, lane STAMP_ARG);
            else if (perLane == 12) rescueWindowScanShort<12>(R, job, windowBase, firstTile, L, tab, present, blockPrefix, mateCodes, firstFlags, ldsBitmap, lane STAMP_ARG);
            else rescueWindowScanShort<16>(R, job, windowBase, firstTile, L, tab, present, blockPrefix, mateCodes, firstFlags, ldsBitmap, lane STAMP_ARG);
        }
        else { rescueWindowScan<false>(R, job, windowBase, firstTile, L, tab, bitmap, lane, pushes); __threadfence(); }
#else
        if (firstTile.codes == 0x123456789abcull) pushes = 1;             // keeps the early window load alive
#endif
        STAMP(3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the set bits in ascending order are the sorted unique candidate list: count them first (an LDS bitmap is one word per lane
        // at most: its words and their running count stay in registers for the enumeration below)
        if (small)
        {
            smallWord = lane < bitmapWords ? ldsBitmap[lane] : 0u;
            smallIncl = waveInclusiveAdd(u32(__popc(smallWord)));
            total = u32(__builtin_amdgcn_readlane(int(smallIncl), 63));
        }
        else for (u32 w0 = 0; w0 < bitmapWords; w0 += 64)
        {
            const u32 w = w0 + lane;
            const u32 word = w < bitmapWords ? __hip_atomic_load(&bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            u32 c = u32(__popc(word));
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            total += c;
        }
    }
    STAMP(4);
#if defined(ISAAC_TIMING_RW_NO_ENUM)
    total = 0;
#endif
    bool fallback = pushes > SHADOW_POSITIONS_MAX;
    if (fallback) total = 0;
    // one allocation per problem from the block's region: the regions keep the atomics on different addresses, and the waves of
    // a block stay independent of each other (no barrier: their windows differ in length)
    u32 candBase = 0xffffffffu;
    if (total)
    {
        if (lane == 0)
        {
            const u32 region = (j / 4) % CAND_REGIONS;
            const u32 at = atomicAdd(rb.candCounter + region, total);
            if (at + total <= rb.candRegionSize) candBase = region * rb.candRegionSize + at;
            else atomicMin(rb.candCounter + CAND_REGIONS + region, at);   // the region is full from here on: these problems fall back
        }
        candBase = __shfl(candBase, 0, 64);
        if (candBase == 0xffffffffu) fallback = true;
    }
    STAMP(6);
    if (!fallback && total)
    {
        const i32 bias = i32(L) - 7;
        u32 running = 0;
        if (small)
        {
            u32 at = candBase + smallIncl - u32(__popc(smallWord));
            while (smallWord)
            {
                const u32 b = u32(__ffs(smallWord)) - 1; smallWord &= smallWord - 1;
                rb.candPositions[at] = i32(lane * 32 + b) - bias; rb.candJob[at] = j; ++at;
            }
        }
        else for (u32 w0 = 0; w0 < bitmapWords; w0 += 64)
        {
            const u32 w = w0 + lane;
            u32 word = w < bitmapWords ? __hip_atomic_load(&bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const u32 c = u32(__popc(word));
            u32 incl = c;
            for (u32 o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            u32 at = candBase + running + incl - c;
            while (word)
            {
                const u32 b = u32(__ffs(word)) - 1; word &= word - 1;
                rb.candPositions[at] = i32(w * 32 + b) - bias; rb.candJob[at] = j; ++at;
            }
            running += __shfl(incl, 63, 64);
        }
    }
    if (lane == 0)
    {
        RescueJob &out = rb.jobs[j];
        out.pushes = pushes; out.fallback = fallback ? 1 : 0; out.candBase = (fallback || !total) ? 0 : candBase; out.nCands = fallback ? 0 : total;
    }
    STAMP(7);
}

__global__ __launch_bounds__(64 * RW_WAVES) void k_rescue_windows(DevParams P, DevReference R, u64 totalBases, const u8 *bcl, u32 clusterBase, RescueBuffers rb)
{
    __shared__ __align__(16) u32 tables[RW_WAVES][RW_TABLE];
    __shared__ u32 ldsBitmaps[RW_WAVES][RW_LDS_BITMAP];
    __shared__ __align__(2048) u32 presentMaps[RW_WAVES][RW_PRESENT_WORDS];      // 2048: rescueWindowScanShort ORs word offsets into the base
    __shared__ __align__(8) u16 blockPrefixes[RW_WAVES][RW_PRESENT_WORDS / 2];     // set bits in front of each 64-bit block of the map
    __shared__ u32 mateWords[RW_WAVES][RW_MATE_WORDS];                               // the mate's bases, 2 bits each
    __shared__ u32 firstFlagWords[RW_WAVES][RW_FLAG_WORDS];                          // one bit per mate position
#if defined(ISAAC_TIMING_RW_EXIT)
    if (clusterBase != 0xffffffffu) return;                                  // timing only: what launching the grid costs
#endif
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // the slot's number in scalar registers: the job record comes through the scalar cache.  (Two or four slots per wavefront, one after
    // the other, to launch fewer workgroups: 3.8 -> 10.0 / 11.7 ms.  The record loads of the second slot may alias the first one's
    // record store, so they become vector loads, the record lives in vector registers and the kernel needs 95 of them.  A resident grid of
    // 8 192 wavefronts over a compacted list of the slots in use, the record's fields moved to scalar registers one by one, has the same
    // 95 registers -- the loop's invariants -- or 64 with spills: 4.7 ms against 3.1 (and that attempt's records differed: a bug that
    // was not chased once the time was known).)
    rescueWindowsProblem(P, R, bcl, clusterBase, rb, blockIdx.x * RW_WAVES + wave, lane, tables[wave], ldsBitmaps[wave], presentMaps[wave], blockPrefixes[wave], mateWords[wave], firstFlagWords[wave]);
}

// With sequencing adapters only (--default-adapters): ShadowAligner::rescueShadow makes a fresh FragmentSequencingAdapterClipper per call and its first candidate
// position initialises the shadow's strand (ShadowAligner.cpp:207,222); the positions of a problem lie ascending from candBase, so that is the first slot.
// A thread per problem; k_rescue_align and the problem's gapped retries clip by what it leaves in the record.
__global__ __launch_bounds__(256) void k_rescue_adapter_ranges(DevParams P, DevReference R, const u8 *bcl, u32 clusterBase, RescueBuffers rb)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= imin(*rb.jobCounter, rb.jobsCap)) return;
    RescueJob &job = rb.jobs[j];
    if (!job.valid || job.fallback || !job.nCands) return;
    const u32 r = job.shadowReadIndex;
    ReadView shadowRead;
    shadowRead.bcl = bcl + u64(clusterBase + job.cluster) * P.clusterLength + P.readOffset[r]; shadowRead.length = P.readLength[r];
    shadowRead.firstCycle = P.firstCycle[r]; shadowRead.endCyclesMasked = 0;
    job.adapterRange = adapterStrandRange(*P.adapters, R, shadowRead, 0 != job.shadowReverse, job.contigId, i64(rb.candPositions[job.candBase]) + job.windowBegin);
}

// (six waves per SIMD: with its candidate in registers the kernel took 87 registers, five waves; held to 80 it has 79 and no scratch: 1.45 -> 1.39 ms, profiles/exp_r5_rescue_align_waves.log)
#ifndef ISAAC_WAVES_RESCUE_ALIGN
#define ISAAC_WAVES_RESCUE_ALIGN 6
#endif
#if ISAAC_WAVES_RESCUE_ALIGN
__attribute__((amdgpu_waves_per_eu(ISAAC_WAVES_RESCUE_ALIGN, ISAAC_WAVES_RESCUE_ALIGN)))
#endif
__global__ __launch_bounds__(256) void k_rescue_align(DevParams P, DevReference Rg, const u8 *bcl, u32 clusterBase, ClusterPools pools, RescueBuffers rb, Counters *counters)
{
    // (one shared copy of the quality tables here: with a copy per lane -- k_align_candidates -- this kernel was slower, 2.31 -> 3.37 ms as it is and 2.66 ->
    // 2.97 ms as a grid that strides over the slots to stage the copies less often; a slot per thread and 94 000 small workgroups stayed the fastest form.
    // profiles/r4_exp_scan_tables*.log)
    ISAAC_STAGE_QUALITY_TABLES(Rg, R)
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    u32 scans = 0;
    // slots in use: below the region's counter and below the first request the region could not serve
    if (i < rb.candCap && i % rb.candRegionSize < imin(imin(rb.candCounter[i / rb.candRegionSize], rb.candCounter[CAND_REGIONS + i / rb.candRegionSize]), rb.candRegionSize))
    {
        const RescueJob &job = rb.jobs[rb.candJob[i]];
        CandSummary summary;
        rescueAlignCandidate(P, R, bcl, clusterBase + job.cluster, pools.meta[job.cluster].endCyclesMasked[job.shadowReadIndex], job, rb.candPositions[i], rb.shadowCands[i], rb.shadowCigars + u64(i) * 3, &summary);
        // the 16 bytes the plan kernels walk.  The tried start and the first CIGAR word are fetched again, through addresses formed here, rather than kept
        // across the scan: the kernel runs at its register cap (six waves per SIMD)
        u32 slot = i; asm volatile("" : "+v"(slot));
        summary.relativePosition = rescueSummaryPosition(*reinterpret_cast<const volatile i32 *>(rb.candPositions + slot), summary.cigarLength, *reinterpret_cast<const volatile u32 *>(rb.shadowCigars + u64(slot) * 3));
        rb.candSummaries[slot] = summary;
        ++scans;
    }
    flushCounter(&Counters::ungappedScans, scans, counters);
}

// one thread per rescue problem: which of its aligned candidates get a gapped retry (ShadowAligner.cpp:232-262).  Problems with
// long candidate lists (repeat families: thousands of entries) would keep one thread walking them long after the rest of the
// grid has finished; they are listed for k_rescue_gapped_plan_long instead.
#ifndef ISAAC_GAPPED_PLAN_LONG
#define ISAAC_GAPPED_PLAN_LONG 48
#endif
static const u32 GAPPED_PLAN_LONG = ISAAC_GAPPED_PLAN_LONG;
__global__ __launch_bounds__(256) void k_rescue_gapped_plan(ClusterPools pools, RescueBuffers rb, GappedBuffers gb, u32 *longList, u32 *longCount, Counters *counters)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nJobs = imin(*rb.jobCounter, rb.jobsCap);
    Counters local; memset(&local, 0, sizeof(local));
    if (j < nJobs && rb.jobs[j].valid && !rb.jobs[j].fallback)
    {
        RescueJob &job = rb.jobs[j];
        // the flat pass's rescue statistics are counted here, one wave reduction instead of one atomic per problem
        ++local.rescueCalls; local.rescueWindowBases += job.windowLen; local.rescueCandidates += job.nCands;
        if (job.nCands > GAPPED_PLAN_LONG) longList[atomicAdd(longCount, 1u)] = j;
        else
        {
            summarizeRescueJob(job, rb.shadowCands, rb.candRank, rb.candSummaries);
            const u32 ecm = pools.meta[job.cluster].endCyclesMasked[job.shadowReadIndex];
            const u32 n = job.nGapped;      // counted by the summary pass; the candidates are walked again only to write the problems
            u32 base = 0;
            if (n)
            {
                base = atomicAdd(gb.counter, n);
                if (base + n > gb.cap) base = 0xffffffffu;     // the cluster's thread runs them itself
                else writeRescueGapped(job, rb.shadowCands, rb.shadowCigars, ecm, gb.jobs + base, rb.candSummaries);
            }
            job.gappedBase = base; job.nGapped = n;
        }
    }
    flushCounters(local, counters);
}

// value of lane `k` (the same k in every lane)
__device__ inline u32 laneValue(u32 v, int k) { return u32(__builtin_amdgcn_readlane(int(v), k)); }
__device__ inline u64 laneValue(u64 v, int k) { return u64(laneValue(u32(v), k)) | (u64(laneValue(u32(v >> 32), k)) << 32); }
__device__ inline double laneValue(double v, int k) { return __longlong_as_double((long long)laneValue(u64(__double_as_longlong(v)), k)); }

// summarizeRescueJob + planRescueGapped (template.h) by one wavefront: 64 candidates are fetched at a time, the statements that
// depend on the order of the list run over register values instead of one memory round trip per candidate.
__global__ __launch_bounds__(256) void k_rescue_gapped_plan_long(ClusterPools pools, RescueBuffers rb, GappedBuffers gb, const u32 *longList, const u32 *longCount)
{
    const u32 lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    const u32 nLong = *longCount;
    const u64 below = (1ull << lane) - 1ull;
    for (u32 t = wave; t < nLong; t += nWaves)
    {
        RescueJob &job = rb.jobs[longList[t]];
        const Cand *cands = rb.shadowCands + job.candBase;
        const CandSummary *summaries = rb.candSummaries + job.candBase;          // the 16 bytes of each candidate the walks look at
        // ---- summarizeRescueJob
        u32 n = 0; i32 best = -1; u32 bestRank = 0, bestMismatches = 0; bool last = false; double bestLp = 0.0;
        u32 nClose = 0; i64 prevPosition = 0; u32 prevMismatches = 0;
        for (u32 c0 = 0; c0 < job.nCands; c0 += 64)
        {
            const u32 c = c0 + lane; const bool in = c < job.nCands;
            double lp = 0.0; i64 position = 0; u32 mismatches = 0; bool aligned = false;
            if (in) { const CandSummary f = summaries[c]; lp = f.logProbability; position = f.relativePosition; mismatches = f.mismatchCount; aligned = 0 != f.cigarLength; }
            const u64 mask = __ballot(aligned);
            if (in) rb.candRank[job.candBase + c] = n + u32(__popcll(mask & below));
            for (u64 m = mask; m; m &= m - 1)
            {
                const int k = __ffsll((long long)m) - 1;
                const double lpk = laneValue(lp, k); const i64 pk = i64(laneValue(u64(position), k)); const u32 mk = laneValue(mismatches, k);
                if (best < 0 || lpLess(bestLp, lpk)) { best = i32(c0) + k; bestRank = n; bestLp = lpk; bestMismatches = mk; }
                if (n && pk - prevPosition < i64(BSW_DISTANCE_CUTOFF) && BSW_MISMATCHES_CUTOFF < prevMismatches) ++nClose;
                prevPosition = pk; prevMismatches = mk; ++n;
            }
            if (c0 + 64 >= job.nCands) last = 0 != ((mask >> (job.nCands - 1 - c0)) & 1);
        }
        const u32 nGapped = (best >= 0 && BSW_MISMATCHES_CUTOFF < bestMismatches) ? nClose : 0;
        // ---- planRescueGapped: every aligned candidate with its aligned predecessor
        u32 base = 0;
        if (nGapped)
        {
            if (0 == lane) base = atomicAdd(gb.counter, nGapped);
            base = laneValue(base, 0);
            if (base + nGapped > gb.cap) base = 0xffffffffu;
        }
        if (nGapped && 0xffffffffu != base)
        {
            const u32 ecm = pools.meta[job.cluster].endCyclesMasked[job.shadowReadIndex];
            u32 emitted = 0; bool havePrev = false; u32 carryC = 0; i64 carryPosition = 0; u32 carryMismatches = 0;
            for (u32 c0 = 0; c0 < job.nCands; c0 += 64)
            {
                const u32 c = c0 + lane; const bool in = c < job.nCands;
                i64 position = 0; u32 mismatches = 0; bool aligned = false;
                if (in) { const CandSummary f = summaries[c]; position = f.relativePosition; mismatches = f.mismatchCount; aligned = 0 != f.cigarLength; }
                const u64 mask = __ballot(aligned);
                const u64 before = mask & below;
                const int pl = before ? 63 - __clzll((long long)before) : 0;
                const i64 shiftedPosition = i64(__shfl(position, pl, 64)); const u32 shiftedMismatches = __shfl(mismatches, pl, 64);
                const bool mine = before ? true : havePrev;
                const u32 prevC = before ? c0 + u32(pl) : carryC;
                const i64 pPosition = before ? shiftedPosition : carryPosition; const u32 pMismatches = before ? shiftedMismatches : carryMismatches;
                const bool flag = aligned && mine && position - pPosition < i64(BSW_DISTANCE_CUTOFF) && BSW_MISMATCHES_CUTOFF < pMismatches;
                const u64 fmask = __ballot(flag);
                if (flag)
                {
                    GappedJob &g = gb.jobs[base + emitted + u32(__popcll(fmask & below))];
                    g.in = cands[prevC]; g.cluster = job.cluster; g.endCyclesMasked = u16(ecm); g.accepted = 0; g.tag = job.candBase + prevC; g.adapterRange = job.adapterRange;
                    g.in.cigarOffset = 0;
                    g.in.position = candUnclippedPosition(g.in, rb.shadowCigars + u64(job.candBase + prevC) * 3); g.in.cigarLength = 0;
                }
                emitted += u32(__popcll(fmask));
                if (mask)
                {
                    const int hi = 63 - __clzll((long long)mask);
                    havePrev = true; carryC = c0 + u32(hi); carryPosition = i64(laneValue(u64(position), hi)); carryMismatches = laneValue(mismatches, hi);
                }
            }
        }
        if (0 == lane)
        {
            job.nAligned = n; job.bestRank = bestRank; job.bestSlot = best < 0 ? 0 : job.candBase + u32(best); job.lastAligned = last ? 1 : 0;
            job.gappedBase = base; job.nGapped = nGapped;
        }
    }
}
