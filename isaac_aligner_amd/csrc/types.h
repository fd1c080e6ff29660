// Shared plain-data layouts of the MI355X seed-and-extend pipeline.
//
// Everything in csrc/*.h is written as ISAAC_HD functions over plain pointers: hipcc compiles them into the gfx950 kernels
// of the product library; tests/hostemu compiles the very same headers with g++ so that the thread-serial device logic can
// be stepped against the oracle in the CPU-only container (a debugging harness, never a product path).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ISAAC_HD __host__ __device__ inline
#else
#define ISAAC_HD inline
#endif

// optional in-kernel section stamps (-DISAAC_KERNEL_STAMPS): shader-clock ticks per section, summed over the sampled waves
// (lane 0 of every wave of every 256th workgroup), printed when the context is destroyed.  A measuring aid, compiled out of
// the product build.  The kernels are in several translation units: each has its own g_stamps and registers a function that prints it
// (isaac_gpu_destroy calls them all).
#if defined(ISAAC_KERNEL_STAMPS) && defined(__HIPCC__)
#include <cstdio>
#include <vector>
static __device__ unsigned long long g_stamps[64];
inline std::vector<void (*)()> &stampPrinters() { static std::vector<void (*)()> v; return v; }
namespace
{
struct StampPrinter
{
    StampPrinter() { stampPrinters().push_back(&print); }
    static void print()
    {
        unsigned long long h[64];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)) != hipSuccess) return;
        for (int i = 0; i < 64; ++i) if (h[i]) fprintf(stderr, "stamp %2d: %llu\n", i, h[i]);
    }
} stampPrinter;
}
#endif
#if defined(ISAAC_KERNEL_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define STAMP_BEGIN() long long stamp_t = clock64()
#define STAMP_PARAM , long long &stamp_t
#define STAMP_ARG , stamp_t
#define STAMP(slot) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 255) == 0) { const long long stamp_n = clock64(); atomicAdd(&g_stamps[slot], (unsigned long long)(stamp_n - stamp_t)); stamp_t = stamp_n; } } while (0)
#else
#define STAMP_BEGIN()
#define STAMP(slot)
#define STAMP_PARAM
#define STAMP_ARG
#endif

namespace isaac
{

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int16_t i16; typedef int32_t i32; typedef int64_t i64;

// ---- reference formats (cited in include/isaac_gpu.h) -------------------------------------------------------------
// ReferencePosition (include/reference/ReferencePosition.hh:51-188): bit 0 neighbors, bits 1..40 position, bits 41..63 contig+1
static const u64 REFPOS_NOMATCH = ((~u64(0)) >> 41) << 41;
static const u32 MAX_CONTIG_ID = u32((~u64(0)) >> 41);
ISAAC_HD u64 refpos(u64 contig, u64 position, bool neighbors = false) { return ((((contig + 1) << 40) | position) << 1) | u64(neighbors); }
ISAAC_HD u32 refposContig(u64 v) { return u32(v >> 41) - 1; }
ISAAC_HD u64 refposPosition(u64 v) { return (v >> 1) & ((u64(1) << 40) - 1); }
ISAAC_HD bool refposIsTooMany(u64 v) { return (v >> 1) == 0; }
ISAAC_HD bool refposIsNoMatch(u64 v) { return v == REFPOS_NOMATCH; }
// SeedId (include/alignment/SeedId.hh:60-127)
ISAAC_HD u64 seedId(u64 tile, u64 barcode, u64 cluster, u64 seed, u64 reverse) { return (tile << 52) | (barcode << 40) | (cluster << 9) | (seed << 1) | reverse; }
ISAAC_HD u32 seedIdSeed(u64 v) { return u32(v >> 1) & 0xff; }
ISAAC_HD u32 seedIdCluster(u64 v) { return u32(v >> 9) & 0x7fffffffu; }
// Cigar (include/alignment/Cigar.hh:52-70,156-168)
enum { OP_ALIGN = 0, OP_INSERT = 1, OP_DELETE = 2, OP_SOFT_CLIP = 4 };
ISAAC_HD u32 cigarOp(u32 len, u32 op) { return (len << 4) | op; }
ISAAC_HD u32 cigarLen(u32 w) { return w >> 4; }
ISAAC_HD u32 cigarCode(u32 w) { u32 c = w & 0xf; return c > 9 ? 9 : c; }

struct Match { u64 seedId; u64 location; };

// ---- parameters as the kernels see them ---------------------------------------------------------------------------
static const u32 MAX_SEEDS = 16;
struct DevSeed { u16 offset, length; u32 readIndex; };
// Sequencing adapters of the flowcell (--default-adapters): flowcell::SequencingAdapterMetadata + the 5-mer position table of
// matchSelector::SequencingAdapter (lib/alignment/matchSelector/SequencingAdapter.cpp:30-56).  Lives in device memory (host memory for the host forms);
// DevParams::adapters is NULL when the list is empty, and every kernel's adapter step is behind that one uniform test.
static const u32 MAX_ADAPTERS = 8, ADAPTER_MATCH_BASES_MIN = 5, MAX_ADAPTER_LENGTH = 126;
struct DevAdapter { u32 length, reverse, clipLength /* 0: unbounded */, pad; char sequence[128]; signed char kmerPositions[1024]; };
struct DevAdapters { u32 n, pad[3]; DevAdapter a[MAX_ADAPTERS]; };
struct Cand;
struct DevParams
{
    i32 gapMatch, gapMismatch, gapOpen, gapExtend, minGapExtend;
    // AlignerBase (lib/alignment/fragmentBuilder/AlignerBase.cpp:32-44)
    u32 normalizedMismatchScore, normalizedGapOpenScore, normalizedGapExtendScore, normalizedMaxGapExtendScore;
    u32 repeatThreshold, gappedMismatchesMax, semialignedGapLimit, baseQualityCutoff;
    u32 ignoreNeighbors, clipSemialigned, clipOverlapping, scatterRepeats;
    i32 dodgyAlignmentScore; u32 mapqThreshold, keepUnaligned; i32 mateDriftRange;
    u32 nReads, readLength[2], readOffset[2], firstCycle[2], clusterLength;
    u32 nSeeds; DevSeed seeds[MAX_SEEDS];
    u32 nPass[2]; u8 passSeeds[2][MAX_SEEDS];  // FindMatchesTransition.cpp:90-110, each list ordered by (read, offset)
    u32 maxSeedsPerRead;
    // sequencing adapters (NULL: none).  Where a read's adapter lies is decided once per read and strand by the first candidate of the list
    // (FragmentSequencingAdapterClipper::checkInitStrand) and used by every later alignment of that strand: adapterRanges holds those four values per cluster
    // (read x strand; begin | end << 16 in strand coordinates, 0: none), addressed by the cluster's first candidate slot relative to adapterCandBase.
    const DevAdapters *adapters;
    u32 *adapterRanges; const Cand *adapterCandBase;
};

// reference::ReferenceKmer<unsigned long> (include/reference/ReferenceKmer.hh:37-54): one entry of the sorted table, one record of a mask file.
// A probe that finds its k-mer has the position in the same 16 bytes (rounds 1-4 kept two arrays: a hit cost a second cache line and a second round trip)
struct TableEntry { u64 kmer, position; };
static_assert(sizeof(TableEntry) == 16, "mask file record");

// reference sequence + sorted k-mer table resident in HBM
struct DevReference
{
    const char *bases;        // all contigs, ASCII ACGTN, concatenated
    u64 totalBases;           // contigOffset[nContigs]
    const u64 *contigOffset;  // n_contigs + 1
    const u8 *contigLoaded;   // MatchSelector.cpp:85-90: contigs without any seed match count as empty
    u32 nContigs;
    const TableEntry *entries;// the sorted 32-mer table (all masks concatenated): k-mer and ReferencePosition side by side, as the mask files hold them
    u64 nKmers;
    const u32 *karyotype;     // contig id translation of ReferenceKmer::getTranslatedPosition, NULL = identity
    const u32 *prefixTable;   // optional: two words per PREFIX_BITS-bit k-mer prefix (+ end sentinel): its first table index, fingerprints of its first four entries
    u32 prefixBits;
    const u32 *packedBases;   // the same bases 2 bits each, 16 per word, base i of a word at bits 2i: A 0, C 1, T 2, G 3 ((ASCII >> 1) & 3)
    const u32 *notBase;       // 1 bit per base: set where it is not one of ACGT (32 per word); both arrays end with spare words
    const double *logMatch;   // Quality::logMatchLookup / logMismatchLookup (lib/alignment/Quality.cpp:34-66), 100 entries each:
    const double *logMismatch;//   entry q at [q * logStride] (1 in global memory; kernels.h: the copies in LDS, one per lane of a half wavefront)
    u32 logStride;
};
ISAAC_HD u64 contigLength(const DevReference &r, u32 contig) { return r.contigOffset[contig + 1] - r.contigOffset[contig]; }

// ---- candidates (alignment::FragmentMetadata reduced to what the path reads) -------------------------------------
static const u16 NON_UNIQUE_NONE = 0xffff;
struct Cand
{
    i64 position;
    double logProbability;
    u32 contigId;
    u32 observedLength;
    u32 smithWatermanScore;
    u32 cigarOffset;      // into the cluster's cigar pool
    u32 alignmentScore;
    u16 cigarLength, mismatchCount, matchesInARow, gapCount, editDistance, lowClipped, highClipped;
    u16 uniqueSeedCount, nonUniqueFirst, nonUniqueSecond, repeatSeedsCount;
    u8 reverse, readIndex; signed char firstSeedIndex; u8 pad;
};
static_assert(sizeof(Cand) == 64, "Cand layout");

ISAAC_HD void candInit(Cand &c, u32 readIndex)
{
    c.position = 0; c.logProbability = 0.0; c.contigId = MAX_CONTIG_ID; c.observedLength = 0; c.smithWatermanScore = 0; c.cigarOffset = 0;
    c.alignmentScore = 0xffffffffu; c.cigarLength = 0; c.mismatchCount = 0; c.matchesInARow = 0; c.gapCount = 0; c.editDistance = 0;
    c.lowClipped = 0; c.highClipped = 0; c.uniqueSeedCount = 0; c.nonUniqueFirst = NON_UNIQUE_NONE; c.nonUniqueSecond = 0; c.repeatSeedsCount = 0;
    c.reverse = 0; c.readIndex = u8(readIndex); c.firstSeedIndex = -1; c.pad = 0;
}
ISAAC_HD bool candAligned(const Cand &c) { return 0 != c.cigarLength; }
ISAAC_HD u32 candObservedLength(const Cand &c) { return candAligned(c) ? c.observedLength : 0; }
ISAAC_HD bool candNoMatch(const Cand &c) { return MAX_CONTIG_ID == c.contigId; }
// FragmentMetadata::isWellAnchored (FragmentMetadata.hh:477-483)
ISAAC_HD bool candWellAnchored(const Cand &c)
{ return c.uniqueSeedCount || (c.nonUniqueFirst != NON_UNIQUE_NONE && c.nonUniqueSecond > c.nonUniqueFirst && u32(c.nonUniqueSecond - c.nonUniqueFirst) >= 32); }
ISAAC_HD void candSetUnaligned(Cand &c) { c.cigarLength = 0; c.alignmentScore = 0xffffffffu; }
ISAAC_HD void candSetNoMatch(Cand &c) { candSetUnaligned(c); c.contigId = MAX_CONTIG_ID; c.position = 0; }
ISAAC_HD void candIncrementClipLeft(Cand &c, u32 bases) { c.position += bases; if (c.reverse) c.highClipped += u16(bases); else c.lowClipped += u16(bases); }
ISAAC_HD void candIncrementClipRight(Cand &c, u32 bases) { if (c.reverse) c.lowClipped += u16(bases); else c.highClipped += u16(bases); }
// FragmentMetadata::operator< / == (FragmentMetadata.hh:419-448)
ISAAC_HD bool candLess(const Cand &a, const Cand &b)
{
    return a.contigId < b.contigId || (a.contigId == b.contigId && (a.position < b.position ||
           (a.position == b.position && (a.reverse < b.reverse || (a.reverse == b.reverse && a.observedLength < b.observedLength)))));
}
ISAAC_HD bool candEqual(const Cand &a, const Cand &b) { return a.position == b.position && a.contigId == b.contigId && a.reverse == b.reverse && a.observedLength == b.observedLength; }

// ---- per-cluster state that lives in HBM between the kernels of one tile -----------------------------------------
// The candidates of a chunk lie back to back in one pool, CSR fashion: a cluster owns as many slots as it has seed matches (every
// candidate starts as a match and the lists only shrink afterwards), at the offset of its first match in the chunk -- no count /
// scan pass is needed, the match offsets are the row pointers.  Read 0's list starts at the cluster's first slot, read 1's behind
// what read 0's held after its first consolidation.  CIGAR words live in one arena per chunk: three words per candidate slot for
// the ungapped alignment at 3 x slot, and extra regions handed out by a bump counter to the few clusters that need more (single
// indels, accepted gapped alignments).  Per cluster 32 bytes of ClusterMeta say where everything is; a typical cluster touches
// 32 + 2.7 x 64 B instead of a fixed 20.5 KB record.
static const u32 CAND_CAP = 128;         // per read (u8 list indexes): 2 strands x seeds/read x (repeatThreshold - 1) = 72 for 4 seeds (2x150), 126 for 7 (2x250)
static const u32 CIGAR_POOL = 1024;      // cigar words of a fixed-capacity ClusterStore (host harness)
static const u32 MATCH_CAP_MAX = 255;       // list indexes are bytes; a cluster with more matches is flagged (CLUSTER_OVERFLOW)
struct ClusterMeta
{
    u32 first;              // first candidate slot (and 3 x first = first cigar word) of the cluster in the chunk's pools
    u32 cigarUsed;          // append cursor of the cluster's cigar words, relative to 3 x first
    u32 cigarCap;           // end of the region the cursor is in, relative to 3 x first
    u16 second, cap;        // read 1's list starts `second` slots in; slots owned
    u16 nCands[2];
    u16 endCyclesMasked[2]; // Read::endCyclesMasked_ after trimLowQualityEnds
    u16 flags;              // CLUSTER_* below
    u8 repeatSeedsCount, built;
    u32 pad;
};
static_assert(sizeof(ClusterMeta) == 32, "ClusterMeta layout");

// What the per-cluster functions work on: a view of the cluster's lists and cigar words (pointers into the pools, or into a
// ClusterStore) plus its small state, kept in registers for the duration of a kernel and written back as ClusterMeta.
struct ClusterFragments
{
    Cand *cands[2];
    u32 candCap[2];         // slots available to each list
    u32 nCands[2];
    u32 cigarUsed;
    u32 cigarCap;
    u32 flags;              // bit0: a fixed-capacity list overflowed
    u32 endCyclesMasked[2];
    u32 repeatSeedsCount;
    u32 built;              // FragmentBuilder::build returned true
    u32 *cigarPool;         // cigarPool[Cand::cigarOffset ...]: the cluster's words
    // the list of read r, r known at run time only: a select between the two pointers.  cands[r] with such an r makes the compiler keep the whole view in
    // scratch memory -- 80 bytes a thread that every thread of a kernel writes on entry: 1.3 GB a launch of k_cluster_sums16, which reached device memory
    // as 1 GB of writes (profiles/r5_final_pmc_summary.json) for one pointer read back.
    ISAAC_HD Cand *list(u32 r) const { return r ? cands[1] : cands[0]; }
    ISAAC_HD u32 listLength(u32 r) const { return r ? nCands[1] : nCands[0]; }
    ISAAC_HD void setListLength(u32 r, u32 n) { if (r) nCands[1] = n; else nCands[0] = n; }
};
ISAAC_HD ClusterFragments clusterView(const ClusterMeta &m, Cand *candPool, u32 *cigarArena)
{
    ClusterFragments f;
    f.cands[0] = candPool + m.first; f.cands[1] = f.cands[0] + m.second;
    f.candCap[0] = m.cap; f.candCap[1] = m.cap - m.second;
    f.nCands[0] = m.nCands[0]; f.nCands[1] = m.nCands[1];
    f.cigarUsed = m.cigarUsed; f.cigarCap = m.cigarCap; f.flags = m.flags;
    f.endCyclesMasked[0] = m.endCyclesMasked[0]; f.endCyclesMasked[1] = m.endCyclesMasked[1];
    f.repeatSeedsCount = m.repeatSeedsCount; f.built = m.built;
    f.cigarPool = cigarArena + 3 * u64(m.first);
    return f;
}
// a fresh view for a cluster that owns `cap` slots from `first` on
ISAAC_HD ClusterFragments clusterViewNew(u32 first, u32 cap, Cand *candPool, u32 *cigarArena)
{
    ClusterMeta m; m.first = first; m.cigarUsed = 0; m.cigarCap = 3 * cap; m.second = 0; m.cap = u16(cap); m.nCands[0] = m.nCands[1] = 0;
    m.endCyclesMasked[0] = m.endCyclesMasked[1] = 0; m.flags = 0; m.repeatSeedsCount = 0; m.built = 0; m.pad = 0;
    return clusterView(m, candPool, cigarArena);
}
ISAAC_HD void clusterViewStore(const ClusterFragments &f, Cand *candPool, ClusterMeta &m)
{
    m.first = u32(f.cands[0] - candPool); m.second = u16(f.cands[1] - f.cands[0]); m.cap = u16(f.candCap[0]);
    m.nCands[0] = u16(f.nCands[0]); m.nCands[1] = u16(f.nCands[1]); m.cigarUsed = f.cigarUsed; m.cigarCap = f.cigarCap; m.flags = u16(f.flags);
    m.endCyclesMasked[0] = u16(f.endCyclesMasked[0]); m.endCyclesMasked[1] = u16(f.endCyclesMasked[1]);
    m.repeatSeedsCount = u8(f.repeatSeedsCount); m.built = u8(f.built); m.pad = 0;
}
// the chunk's pools as the kernels receive them
struct ClusterPools { ClusterMeta *meta; Cand *cands; u32 *cigars; u32 candCap /* slots */; u32 cigarCap /* words */; u32 *cigarNext /* bump counter of the extra regions */;
                      u32 *shortFlag /* set when a cluster's slots lie beyond candCap */; };
// `words` more cigar words for a cluster whose cursor region is full or not the right size: a fresh region from the arena's
// bump counter; the cluster's earlier words stay where they are (offsets are relative to its first word).  No room: the view's
// capacity stays as it is and the pool's own overflow flag does the rest.
ISAAC_HD void clusterCigarExtra(ClusterFragments &f, u32 *cigarArena, u32 at, u32 words, u32 arenaCap)
{
    if (u64(at) + words > arenaCap) return;
    const u32 base = u32(f.cigarPool - cigarArena);
    f.cigarUsed = at - base; f.cigarCap = f.cigarUsed + words;
}
// fixed-capacity backing of one cluster's view (tests/hostemu; the serial forms)
struct ClusterStore
{
    Cand cands[2 * CAND_CAP]; u32 cigarPool[CIGAR_POOL];
    ISAAC_HD ClusterFragments view() { ClusterFragments f = clusterViewNew(0, 2 * CAND_CAP, cands, cigarPool); f.cigarCap = CIGAR_POOL; return f; }
};
enum { CLUSTER_OVERFLOW = 1,
       CLUSTER_INDEL_PENDING = 2,     // << read index (bits 1, 2)
       CLUSTER_ALIGN_PENDING = 8 };   // the flat alignment list was full: k_finish_candidates runs the cluster's ungapped scans itself   // << read index: the single-indel stage of this read is still to run (finishSimpleIndels)

struct Counters
{
    u64 clusters, probes, probeSteps, matches, candidates, ungappedScans, bswJobs, bswAccepted, simpleIndels,
        rescueCalls, rescueWindowBases, rescueCandidates, rescueBsw, overflowClusters, mapqNearInteger, heavyClusters,
        // why clusters went to the wave-per-cluster pass (k_cluster_sums) and how many took the workgroup-per-cluster sums
        residualCapacity, residualNearTie, residualOversize, largeSums;
};

// per-cluster facts TemplateLengthDistribution::addTemplate looks at (TemplateLengthStatistics.cpp:275-314)
struct TlsSample { u32 n0, n1; u32 contig0, contig1; i64 pos0, pos1; u32 obs0, obs1; u8 rev0, rev1; u8 insertEnd; u8 valid; };

// Quality.hh:104-112
ISAAC_HD bool lpEquals(double l, double r) { double d = l - r; if (d < 0) d = -d; return 0.0000001 >= d; }
ISAAC_HD bool lpLess(double l, double r) { return !lpEquals(l, r) && l < r; }

template <typename T> ISAAC_HD T imin(T a, T b) { return a < b ? a : b; }
template <typename T> ISAAC_HD T imax(T a, T b) { return a > b ? a : b; }

} // namespace isaac
