// Device side of the sorted-reference builder (isaac-sort-reference: lib/reference/ReferenceSorter.cpp:105-261 and
// lib/reference/NeighborsFinder.cpp:193-446), sized for a human genome on one MI355X.
//
// The reference sorts one mask (the top 6 bits of the 32-mer, --mask-width 6) per process and concatenates the 64 mask files; the
// table built here is the same sequence.  One pass counts, per workgroup of INDEX_TILE positions, how many k-mers (both strands)
// fall into each of the 64 masks; then every mask is emitted in position order, radix-sorted (stable: equal k-mers stay in
// position order, forward strand first), analysed for repeats and appended to the resident table.  All indices that can exceed
// 2^32 (positions in the genome, entries of the table, distinct k-mers) are 64-bit.
//
// Neighbour annotation: a k-mer has neighbours when another distinct k-mer of the reference (either strand) differs from it in
// 1..4 bases.  As in NeighborsFinder, every pair within 4 mismatches agrees on 4 of the 8 four-base blocks, so for each of the
// C(8,4) = 70 choices the distinct k-mers are grouped by those 16 bases and compared on the other 16.  The k-mers travel as
// byte-shuffled 64-bit keys (chosen blocks in the low 32 bits) with their flag as the sort payload; only those 32 bits are
// sorted, and a k-mer that is already flagged is not compared again.
#pragma once
#include "types.h"

namespace isaac
{

static const u32 INDEX_MASKS = 64, INDEX_MASK_BITS = 6;
static const u32 INDEX_THREADS = 256, INDEX_PER_THREAD = 16, INDEX_TILE = INDEX_THREADS * INDEX_PER_THREAD;

#if defined(__HIPCC__)

// The bases of positions [g, g + 16 + 31] for a thread whose first position g is a multiple of 16: 2-bit codes and not-ACGT flags
struct KmerWindow
{
    u64 lo, hi;      // stream bits [2g, 2g + 128): base g + k at bits 2k
    u64 notBase;     // bit k: base g + k is not one of ACGT
    __device__ inline void load(const u32 *packed, const u32 *notBaseBits, u64 g)
    {
        const u32 *pw = packed + (g >> 4);
        lo = u64(pw[0]) | (u64(pw[1]) << 32); hi = u64(pw[2]) | (u64(pw[3]) << 32);
        const u32 *pn = notBaseBits + (g >> 5);
        notBase = (u64(pn[0]) | (u64(pn[1]) << 32)) >> (g & 31);        // 48 bits wanted, g & 31 is 0 or 16
    }
    // the 32 bases starting at g + k, base j of the k-mer at bits 2j, in the reference's coding (A 0, C 1, G 2, T 3)
    __device__ inline u64 bases(u32 k) const
    {
        u64 v = k ? (lo >> (2 * k)) | (hi << (64 - 2 * k)) : lo;
        return v ^ ((v >> 1) & 0x5555555555555555ull);                      // packed copy: T 2, G 3
    }
    __device__ inline bool acgt(u32 k) const { return 0 == u32(notBase >> k); }
};
// first base in the most significant bits (oligo::KmerGenerator order)
__device__ inline u64 forwardKmer(u64 basesLsbFirst)
{
    const u64 r = __brevll(basesLsbFirst);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}
// reverse complement, first base in the most significant bits: the complement of every base, in the order they were read
__device__ inline u64 reverseKmer(u64 basesLsbFirst) { return ~basesLsbFirst; }

// contig of global position p (contigOffset has nContigs + 1 entries)
__device__ inline u32 contigOf(const u64 *contigOffset, u32 nContigs, u64 p)
{
    u32 lo = 0, hi = nContigs;
    while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (contigOffset[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

// one thread's 16 positions; f(k, contig, positionInContig, fwd, rc) for every position that starts a 32-mer
template <typename F>
__device__ inline void forEachKmer(const u32 *packed, const u32 *notBaseBits, const u64 *contigOffset, u32 nContigs, u64 totalBases, u64 g, F f)
{
    if (g >= totalBases) return;
    KmerWindow w; w.load(packed, notBaseBits, g);
    u32 contig = contigOf(contigOffset, nContigs, g);
    u64 contigEnd = contigOffset[contig + 1];
#pragma unroll
    for (u32 k = 0; k < INDEX_PER_THREAD; ++k)
    {
        const u64 p = g + k;
        if (p >= totalBases) break;
        while (p >= contigEnd) contigEnd = contigOffset[++contig + 1];
        if (p + 32 > contigEnd || !w.acgt(k)) continue;
        const u64 b = w.bases(k);
        f(k, contig, p - contigOffset[contig], forwardKmer(b), reverseKmer(b));
    }
}

// counts[mask * nBlocks + block]: k-mers (both strands) of the block's positions whose top 6 bits are `mask`; validCount: positions
// that start a 32-mer
__global__ __launch_bounds__(INDEX_THREADS) void k_index_count(const u32 *packed, const u32 *notBaseBits, const u64 *contigOffset, u32 nContigs, u64 totalBases,
                                                              u64 nBlocks, u32 *counts, unsigned long long *validCount)
{
    __shared__ u32 hist[INDEX_MASKS];
    __shared__ u32 valid;
    if (threadIdx.x < INDEX_MASKS) hist[threadIdx.x] = 0;
    if (0 == threadIdx.x) valid = 0;
    __syncthreads();
    const u64 g = (u64(blockIdx.x) * INDEX_THREADS + threadIdx.x) * INDEX_PER_THREAD;
    u32 mine = 0;
    forEachKmer(packed, notBaseBits, contigOffset, nContigs, totalBases, g, [&](u32, u32, u64, u64 fwd, u64 rc)
    { atomicAdd(&hist[fwd >> (64 - INDEX_MASK_BITS)], 1u); atomicAdd(&hist[rc >> (64 - INDEX_MASK_BITS)], 1u); ++mine; });
    if (mine) atomicAdd(&valid, mine);
    __syncthreads();
    if (threadIdx.x < INDEX_MASKS) counts[u64(threadIdx.x) * nBlocks + blockIdx.x] = hist[threadIdx.x];
    if (0 == threadIdx.x && valid) atomicAdd(validCount, (unsigned long long)valid);
}

// the k-mers of one mask in position order (forward strand first): keys = k-mer, vals = ReferencePosition value with bit 0 set
// for reverse-complement occurrences (they only take part in the repeat count, ReferenceSorter.cpp:179-222)
__global__ __launch_bounds__(INDEX_THREADS) void k_index_emit(const u32 *packed, const u32 *notBaseBits, const u64 *contigOffset, u32 nContigs, u64 totalBases,
                                                             u32 mask, const u64 *blockBase, u64 *keys, u64 *vals)
{
    __shared__ u32 waveTotals[INDEX_THREADS / 64];
    const u64 g = (u64(blockIdx.x) * INDEX_THREADS + threadIdx.x) * INDEX_PER_THREAD;
    u32 mine = 0;
    forEachKmer(packed, notBaseBits, contigOffset, nContigs, totalBases, g, [&](u32, u32, u64, u64 fwd, u64 rc)
    { mine += u32((fwd >> (64 - INDEX_MASK_BITS)) == mask) + u32((rc >> (64 - INDEX_MASK_BITS)) == mask); });
    // exclusive prefix of `mine` over the block
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = mine;
    for (u32 o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (63 == lane) waveTotals[wave] = incl;
    __syncthreads();
    u32 before = 0;
    for (u32 w = 0; w < wave; ++w) before += waveTotals[w];
    if (!mine) return;
    u64 at = blockBase[blockIdx.x] + before + incl - mine;
    forEachKmer(packed, notBaseBits, contigOffset, nContigs, totalBases, g, [&](u32, u32 contig, u64 position, u64 fwd, u64 rc)
    {
        if ((fwd >> (64 - INDEX_MASK_BITS)) == mask) { keys[at] = fwd; vals[at] = refpos(contig, position, false); ++at; }
        if ((rc >> (64 - INDEX_MASK_BITS)) == mask) { keys[at] = rc; vals[at] = refpos(contig, position, true); ++at; }
    });
}

// u32 counts of one mask -> u64 (the running sum over a genome's worth of blocks does not fit 32 bits in general)
__global__ void k_index_widen(const u32 *in, u64 n, u64 *out)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// ---- repeat analysis of one sorted mask (n < 2^31 elements) -----------------------------------------------------
__global__ void k_run_heads(const u64 *keys, u64 n, u32 *head, u32 *isFwd, const u64 *vals)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    isFwd[i] = (vals[i] & 1) ? 0 : 1;
}
// runId = inclusive scan of head - 1; per run: start index, total count, forward count (the last element of a run writes the
// totals using the prefix sums)
__global__ void k_run_totals(const u64 *keys, u64 n, const u32 *runIdIncl, const u32 *fwdExcl, const u32 *isFwd, u32 *runStart, u32 *runTotal, u32 *runFwd)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 run = runIdIncl[i] - 1;
    if (i == 0 || keys[i] != keys[i - 1]) runStart[run] = u32(i);
    if (i + 1 == n || keys[i] != keys[i + 1])
    {
        runTotal[run] = u32(i);                 // for now the index of the last element
        runFwd[run] = fwdExcl[i] + isFwd[i];    // for now the inclusive forward prefix at the end
    }
}
__global__ void k_run_finish(u32 nRuns, const u32 *fwdExcl, const u32 *runStart, u32 *runTotal, u32 *runFwd)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRuns) return;
    const u32 start = runStart[r];
    runTotal[r] = runTotal[r] - start + 1;
    runFwd[r] = runFwd[r] - fwdExcl[start];
}
__global__ void k_emit_flags(u64 n, const u32 *runIdIncl, const u32 *isFwd, const u32 *runStart, const u32 *runTotal, const u32 *runFwd, u32 repeatThreshold, u32 *emit)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 run = runIdIncl[i] - 1;
    u32 e = 0;
    if (runFwd[run])
    {
        if (repeatThreshold < runTotal[run]) e = (runStart[run] == i) ? 1 : 0;   // a single TooManyMatch entry (ReferenceSorter.cpp:201-222)
        else e = isFwd[i];
    }
    emit[i] = e;
}
// entries of the mask appended to the table at outBase; entryRun: the run (distinct k-mer of the mask) an entry belongs to
__global__ void k_emit_entries(const u64 *keys, const u64 *vals, u64 n, const u32 *runIdIncl, const u32 *runTotal, u32 repeatThreshold, const u32 *emit, const u32 *emitSlot,
                               u64 outBase, TableEntry *out, u32 *entryRun)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n || !emit[i]) return;
    const u32 run = runIdIncl[i] - 1;
    const u64 s = outBase + emitSlot[i];
    TableEntry e; e.kmer = keys[i]; e.position = (repeatThreshold < runTotal[run]) ? 0 : (vals[i] & ~u64(1));
    out[s] = e;
    if (entryRun) entryRun[s] = run;
}
// distinct k-mers of the mask (both strands) = run heads, appended at distinctBase
__global__ void k_distinct(const u64 *keys, u64 n, const u32 *head, const u32 *runIdIncl, u64 distinctBase, u64 *distinct)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    distinct[distinctBase + runIdIncl[i] - 1] = keys[i];
}

// ---- neighbour annotation ------------------------------------------------------------------------------------------
// out byte j = in byte sel[j] (sel packed 4 bits per byte, byte 0 = least significant)
struct ByteShuffle { u32 selLo, selHi; };   // v_perm_b32 selectors over {hi word, lo word} of the key
// (grid-stride: a human genome has 5.8 G distinct k-mers, and a launch holds fewer than 2^32 work-items)
__global__ void k_shuffle_keys(u64 *keys, u64 n, ByteShuffle s)
{
    for (u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += u64(gridDim.x) * blockDim.x)
    {
        const u64 v = keys[i];
        const u32 lo = __builtin_amdgcn_perm(u32(v >> 32), u32(v), s.selLo), hi = __builtin_amdgcn_perm(u32(v >> 32), u32(v), s.selHi);
        keys[i] = u64(lo) | (u64(hi) << 32);
    }
}
__device__ inline u32 baseDistance32(u32 a, u32 b) { u32 x = a ^ b; x = (x | (x >> 1)) & 0x55555555u; return u32(__popc(x)); }
// NeighborsFinder::markNeighbors (:395-446) for keys grouped by their LOW 32 bits (the chosen blocks; rocPRIM 4.2 missorts small
// inputs when the sorted bit range does not start at bit 0, so the group key sits there): element i is flagged when another
// element of its group differs from it in 1..4 of the 16 bases of the high half.  Every element looks for itself and stops at
// the first hit.
__global__ void k_mark_neighbors(const u64 *keys, u8 *flags, u64 n)
{
    for (u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += u64(gridDim.x) * blockDim.x)
    {
        if (flags[i]) continue;
        const u64 mine = keys[i];
        const u32 group = u32(mine), rest = u32(mine >> 32);
        bool found = false;
        for (u64 j = i + 1; j < n && !found; ++j)
        {
            const u64 o = keys[j];
            if (u32(o) != group) break;
            const u32 d = baseDistance32(rest, u32(o >> 32));
            found = d && d <= 4;
        }
        for (u64 j = i; !found && j-- > 0;)
        {
            const u64 o = keys[j];
            if (u32(o) != group) break;
            const u32 d = baseDistance32(rest, u32(o >> 32));
            found = d && d <= 4;
        }
        if (found) flags[i] = 1;
    }
}
// table entries take the flag of their k-mer: entry i of mask m belongs to distinct k-mer distinctBase[m] + entryRun[i]
__global__ void k_apply_neighbors(TableEntry *entries, const u32 *entryRun, u64 first, u64 n, u64 distinctBase, const u8 *flags)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 e = first + i;
    const u64 p = entries[e].position;
    if (refposIsTooMany(p)) return;
    if (flags[distinctBase + entryRun[e]]) entries[e].position = p | 1;
}

// ---- mask-file records are the resident table's entries: a streamed table only has its order checked ----------------------------------
typedef TableEntry ReferenceKmerRecord;
// entries [at, at + n) of a table that is being streamed in: ascending k-mers, also across the boundary to the piece before
__global__ void k_check_order(const TableEntry *entries, u64 at, u64 n, u32 *disorder)
{
    const u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 e = at + i;
    if (e && entries[e - 1].kmer > entries[e].kmer) *disorder = 1;
}

#endif // __HIPCC__

} // namespace isaac
