// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Plain C entry points over the restatement so that tests/, smoke() and bench.py's cpu_baseline leg can drive it with ctypes.
// The record layouts mirror include/isaac_gpu.h byte for byte (checked by tests/test_abi.py) but are declared independently.
#include "oracle.hpp"
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <mutex>
#include <memory>
#include <atomic>
#include <algorithm>

using namespace oracle;

extern "C" {

typedef struct
{
    int32_t gap_match, gap_mismatch, gap_open, gap_extend, min_gap_extend;
    uint32_t repeat_threshold, gapped_mismatches_max, semialigned_gap_limit, base_quality_cutoff;
    uint32_t ignore_neighbors, clip_semialigned, clip_overlapping, scatter_repeats;
    int32_t dodgy_alignment_score; uint32_t mapq_threshold; uint32_t keep_unaligned; int32_t mate_drift_range;
    uint32_t first_pass_seeds, seed_length;
    uint32_t n_reads; uint32_t read_length[2];
    uint32_t n_seeds;
    struct { uint16_t offset, length; uint32_t read_index; } seeds[16];
    uint32_t n_adapters;
    struct { char sequence[128]; uint32_t reverse, clip_length; } adapters[8];
} oracle_params;

typedef struct { uint32_t min, max, median, low_std_dev, high_std_dev; int32_t best_model[2]; uint32_t stable, mate_min, mate_max; } oracle_tls;

typedef struct
{
    int64_t position; double log_probability;
    uint32_t cluster, read_index, contig_id, observed_length, reverse, mismatch_count, matches_in_a_row, gap_count, edit_distance,
             smith_waterman_score, unique_seed_count, non_unique_first, non_unique_second, repeat_seeds_count, cigar_offset, cigar_length,
             low_clipped, high_clipped;
    int32_t first_seed_index; uint32_t reserved;
} oracle_candidate;

static thread_local std::string g_error;
const char *oracle_last_error() { return g_error.c_str(); }

static Params toParams(const oracle_params *c)
{
    Params p;
    p.gapMatchScore = c->gap_match; p.gapMismatchScore = c->gap_mismatch; p.gapOpenScore = c->gap_open; p.gapExtendScore = c->gap_extend; p.minGapExtendScore = c->min_gap_extend;
    p.repeatThreshold = c->repeat_threshold; p.gappedMismatchesMax = c->gapped_mismatches_max; p.semialignedGapLimit = c->semialigned_gap_limit;
    p.baseQualityCutoff = c->base_quality_cutoff; p.ignoreNeighbors = c->ignore_neighbors; p.clipSemialigned = c->clip_semialigned; p.clipOverlapping = c->clip_overlapping;
    p.scatterRepeats = c->scatter_repeats; p.dodgyAlignmentScore = c->dodgy_alignment_score; p.mapqThreshold = c->mapq_threshold; p.keepUnaligned = c->keep_unaligned;
    p.mateDriftRange = c->mate_drift_range; p.firstPassSeeds = c->first_pass_seeds; p.seedLength = c->seed_length;
    unsigned offset = 0, firstCycle = 1;
    for (unsigned r = 0; r < c->n_reads; ++r)
    {
        ReadMetadata rm = { c->read_length[r], r, offset, firstCycle }; p.reads.push_back(rm);
        offset += c->read_length[r]; firstCycle += c->read_length[r];
    }
    for (unsigned s = 0; s < c->n_seeds; ++s) { SeedMetadata sm = { c->seeds[s].offset, c->seeds[s].length, c->seeds[s].read_index, s }; p.seeds.push_back(sm); }
    if (c->n_adapters > 8) throw std::runtime_error("at most 8 adapters");
    for (unsigned a = 0; a < c->n_adapters; ++a)
    {
        SequencingAdapterMetadata m; m.sequence.assign(c->adapters[a].sequence, strnlen(c->adapters[a].sequence, sizeof(c->adapters[a].sequence)));
        m.reverse = 0 != c->adapters[a].reverse; m.clipLength = c->adapters[a].clip_length;
        p.adapters.push_back(m);
    }
    return p;
}

// default parameters for the given read geometry (AlignOptions.cpp:77-160 + "auto" seeds)
int oracle_default_params(uint32_t n_reads, uint32_t len1, uint32_t len2, oracle_params *out)
{
    try
    {
        const Params p = makeParams(n_reads, len1, len2);
        memset(out, 0, sizeof(*out));
        out->gap_match = p.gapMatchScore; out->gap_mismatch = p.gapMismatchScore; out->gap_open = p.gapOpenScore; out->gap_extend = p.gapExtendScore; out->min_gap_extend = p.minGapExtendScore;
        out->repeat_threshold = p.repeatThreshold; out->gapped_mismatches_max = p.gappedMismatchesMax; out->semialigned_gap_limit = p.semialignedGapLimit;
        out->base_quality_cutoff = p.baseQualityCutoff; out->ignore_neighbors = p.ignoreNeighbors; out->clip_semialigned = p.clipSemialigned; out->clip_overlapping = p.clipOverlapping;
        out->scatter_repeats = p.scatterRepeats; out->dodgy_alignment_score = p.dodgyAlignmentScore; out->mapq_threshold = p.mapqThreshold; out->keep_unaligned = p.keepUnaligned;
        out->mate_drift_range = p.mateDriftRange; out->first_pass_seeds = p.firstPassSeeds; out->seed_length = p.seedLength;
        out->n_reads = n_reads; out->read_length[0] = len1; out->read_length[1] = n_reads > 1 ? len2 : 0;
        if (p.seeds.size() > 16) throw std::runtime_error("too many seeds");
        out->n_seeds = uint32_t(p.seeds.size());
        for (size_t s = 0; s < p.seeds.size(); ++s) { out->seeds[s].offset = uint16_t(p.seeds[s].offset); out->seeds[s].length = uint16_t(p.seeds[s].length); out->seeds[s].read_index = p.seeds[s].readIndex; }
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- banded smith-waterman leaf (BandedSmithWaterman.hh:75-86)
int oracle_bsw_align(int match, int mismatch, int gap_open, int gap_extend, int max_read_length,
                     const char *query, uint32_t query_length, const char *database /* query_length + 15 bytes */,
                     uint32_t *cigar_out, uint32_t cigar_capacity, uint32_t *n_ops, uint32_t *offset)
{
    try
    {
        BandedSmithWaterman bsw(match, mismatch, gap_open, gap_extend, max_read_length);
        Cigar cigar;
        *offset = bsw.align(query, query + query_length, database, database + query_length + 15, cigar);
        *n_ops = uint32_t(cigar.size());
        if (cigar.size() > cigar_capacity) throw std::runtime_error("cigar capacity");
        memcpy(cigar_out, cigar.data(), cigar.size() * 4);
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
// this thread's banded Smith-Waterman rows lane by lane (1) or sixteen lanes at a time (0, where the build has AVX2): the check of the one against the other
void oracle_bsw_force_scalar(int on) { bswForceScalarRows(on); }
// constructor overflow rule only (BandedSmithWaterman.cpp:47-53): returns 1 when the reference would throw
int oracle_bsw_check(int match, int mismatch, int gap_open, int gap_extend, int max_read_length)
{
    try { BandedSmithWaterman bsw(match, mismatch, gap_open, gap_extend, max_read_length); return 0; }
    catch (const std::exception &) { return 1; }
}

// ---- reference handle: contigs + sorted index
struct oracle_ref
{
    ContigList contigs; SortedReference index; std::vector<uint8_t> contigHasMatches;
    // the list a call works on (contigs without matches emptied), kept from call to call: a copy of a human genome per call was a second and a half of
    // what bench.py's cpu_baseline timed as "selection"
    std::mutex filteredLock; std::vector<uint8_t> filteredKey; std::shared_ptr<const ContigList> filtered;
};

oracle_ref *oracle_ref_create(const char *bases, const uint64_t *offsets /* n+1 */, uint32_t n_contigs)
{
    oracle_ref *r = new oracle_ref;
    for (uint32_t c = 0; c < n_contigs; ++c)
    {
        Contig contig; contig.index = c; contig.name = "c" + std::to_string(c);
        contig.forward.assign(bases + offsets[c], bases + offsets[c + 1]);
        r->contigs.push_back(contig);
        r->index.karyotype.push_back(c);
    }
    r->contigHasMatches.assign(n_contigs, 0);
    return r;
}
void oracle_ref_destroy(oracle_ref *r) { delete r; }
int oracle_ref_build_index(oracle_ref *r, uint32_t repeat_threshold, int annotate_neighbors, uint32_t neighborhood_width)
{
    try { r->index = buildSortedReference(r->contigs, 32, repeat_threshold, annotate_neighbors != 0, neighborhood_width); return 0; }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
int oracle_ref_build_index_mt(oracle_ref *r, uint32_t repeat_threshold, int annotate_neighbors, uint32_t neighborhood_width, uint32_t n_threads)
{
    try { r->index = buildSortedReference(r->contigs, 32, repeat_threshold, annotate_neighbors != 0, neighborhood_width, n_threads); return 0; }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
// load an index in mask-file layout ({u64 kmer, u64 position} sorted by kmer)
void oracle_ref_set_index(oracle_ref *r, const uint64_t *kmer_pos_pairs, uint64_t n)
{
    r->index.kmers.resize(n);
    memcpy(r->index.kmers.data(), kmer_pos_pairs, n * 16);
}
// SortedReferenceMetadata::Contig::karyotypeIndex_ of every stored contig index (MatchFinder.cpp:51-66)
void oracle_ref_set_karyotype(oracle_ref *r, const uint32_t *karyotype, uint32_t n) { r->index.karyotype.assign(karyotype, karyotype + n); }
uint64_t oracle_ref_index_size(const oracle_ref *r) { return r->index.kmers.size(); }
void oracle_ref_get_index(const oracle_ref *r, uint64_t *kmer_pos_pairs) { memcpy(kmer_pos_pairs, r->index.kmers.data(), r->index.kmers.size() * 16); }

// ---- find matches for one tile (both seed passes), sorted by (cluster, location, seed, reverse)
int oracle_find_matches(oracle_ref *r, const oracle_params *cp, const uint8_t *bcl, uint32_t n_clusters, uint32_t tile,
                        uint64_t *matches_out /* pairs {seedId, location} */, uint64_t capacity, uint64_t *n_out, uint8_t *contig_has_matches)
{
    try
    {
        const Params p = toParams(cp);
        std::vector<Match> matches;
        std::vector<uint8_t> hits(r->contigs.size(), 0);
        findTileMatches(p, r->index, bcl, n_clusters, tile, matches, hits);
        *n_out = matches.size();
        if (matches.size() > capacity) throw std::runtime_error("match capacity");
        memcpy(matches_out, matches.data(), matches.size() * 16);
        for (size_t i = 0; i < hits.size(); ++i) { if (contig_has_matches) contig_has_matches[i] |= hits[i]; }
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

void oracle_set_lookup_mode(int mode) { setLookupMode(mode); }

// The same on n_threads host threads, for the CPU baseline of bench.py: the tile is cut into contiguous cluster ranges, every
// thread runs the unchanged single-tile function (its own merge-join over the whole index, as every mask thread of the
// reference streams whole mask files) on its range, and the cluster numbers are put back into the SeedIds afterwards.
int oracle_find_matches_mt(oracle_ref *r, const oracle_params *cp, const uint8_t *bcl, uint32_t n_clusters, uint32_t tile, uint32_t n_threads,
                           uint64_t *matches_out, uint64_t capacity, uint64_t *n_out, uint8_t *contig_has_matches)
{
    try
    {
        const Params p = toParams(cp);
        std::vector<Match> matches;
        std::vector<uint8_t> hits(r->contigs.size(), 0);
        findTileMatchesParallel(p, r->index, bcl, n_clusters, tile, n_threads ? n_threads : 1, matches, hits);
        if (matches.size() > capacity) throw std::runtime_error("match capacity");
        memcpy(matches_out, matches.data(), matches.size() * 16);
        *n_out = matches.size();
        for (size_t i = 0; i < hits.size(); ++i) if (contig_has_matches) contig_has_matches[i] |= hits[i];
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

static std::shared_ptr<const ContigList> filteredContigsShared(oracle_ref *r, const uint8_t *contig_loaded)
{
    // MatchSelector.cpp:85-90,138: contigs without any match are not loaded (empty sequence, length 0)
    std::vector<uint8_t> key(r->contigs.size(), 1);
    if (contig_loaded) key.assign(contig_loaded, contig_loaded + r->contigs.size());
    std::lock_guard<std::mutex> hold(r->filteredLock);
    if (!r->filtered || key != r->filteredKey)
    {
        std::shared_ptr<ContigList> c(new ContigList(r->contigs));
        for (size_t i = 0; i < c->size(); ++i) if (!key[i]) (*c)[i].forward.clear();
        r->filtered = c; r->filteredKey = key;
    }
    return r->filtered;
}

static void fillCandidate(oracle_candidate &o, const FragmentMetadata &f, uint32_t cluster, uint32_t cigarOffset)
{
    memset(&o, 0, sizeof(o));
    o.position = f.position; o.log_probability = f.logProbability; o.cluster = cluster; o.read_index = f.readIndex; o.contig_id = f.contigId;
    o.observed_length = f.observedLength; o.reverse = f.reverse; o.mismatch_count = f.mismatchCount; o.matches_in_a_row = f.matchesInARow; o.gap_count = f.gapCount;
    o.edit_distance = f.editDistance; o.smith_waterman_score = f.smithWatermanScore; o.unique_seed_count = f.uniqueSeedCount;
    o.non_unique_first = f.nonUniqueSeedOffsets.first; o.non_unique_second = f.nonUniqueSeedOffsets.second; o.repeat_seeds_count = f.repeatSeedsCount;
    o.cigar_offset = cigarOffset; o.cigar_length = f.cigarLength; o.low_clipped = f.lowClipped; o.high_clipped = f.highClipped; o.first_seed_index = f.firstSeedIndex;
}

// ---- FragmentBuilder::build for every cluster of the match list (FragmentBuilder.hh:62-70); candidates in (cluster, read, list order)
int oracle_build_fragments(oracle_ref *r, const oracle_params *cp, const uint8_t *contig_loaded, const uint8_t *bcl, uint32_t tile,
                           const uint64_t *matches, uint64_t n_matches, int with_gaps, int trim,
                           oracle_candidate *out, uint64_t capacity, uint64_t *n_out, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar)
{
    try
    {
        const Params p = toParams(cp);
        const std::shared_ptr<const ContigList> contigsHeld = filteredContigsShared(r, contig_loaded); const ContigList &contigs = *contigsHeld;
        FragmentBuilder fb(p);
        Cluster cluster;
        const Match *mb = reinterpret_cast<const Match *>(matches), *me = mb + n_matches;
        uint64_t n = 0, nc = 0;
        const unsigned clusterLength = p.clusterLength();
        while (mb != me)
        {
            const uint64_t clusterId = SeedId(mb->seedId).getCluster();
            const Match *mn = mb; while (mn != me && SeedId(mn->seedId).getCluster() == clusterId) ++mn;
            if (!ReferencePosition::fromValue(mb->location).isNoMatch())
            {
                cluster.init(p.reads, bcl + clusterId * clusterLength, tile, clusterId, true);
                if (trim) trimLowQualityEnds(cluster, p.baseQualityCutoff);
                fb.build(contigs, p.reads, p.seeds, mb, mn, cluster, with_gaps != 0);
                for (unsigned rd = 0; rd < 2; ++rd) for (size_t i = 0; i < fb.fragments[rd].size(); ++i)
                {
                    const FragmentMetadata &f = fb.fragments[rd][i];
                    if (n >= capacity || nc + f.cigarLength > cigar_capacity) throw std::runtime_error("candidate capacity");
                    fillCandidate(out[n], f, uint32_t(clusterId), uint32_t(nc));
                    if (f.cigarLength) memcpy(cigar_out + nc, f.cigarBuffer->data() + f.cigarOffset, f.cigarLength * 4);
                    nc += f.cigarLength; ++n;
                }
            }
            mb = mn;
        }
        *n_out = n; *n_cigar = nc;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

static void toTls(const TemplateLengthStatistics &t, oracle_tls *o)
{
    o->min = t.min; o->max = t.max; o->median = t.median; o->low_std_dev = t.lowStdDev; o->high_std_dev = t.highStdDev;
    o->best_model[0] = t.bestModels[0]; o->best_model[1] = t.bestModels[1]; o->stable = t.stable; o->mate_min = t.mateMin; o->mate_max = t.mateMax;
}
static TemplateLengthStatistics fromTls(const oracle_tls *o)
{
    TemplateLengthStatistics t;
    t.min = o->min; t.max = o->max; t.median = o->median; t.lowStdDev = o->low_std_dev; t.highStdDev = o->high_std_dev;
    t.bestModels[0] = TemplateLengthStatistics::AlignmentModel(o->best_model[0]); t.bestModels[1] = TemplateLengthStatistics::AlignmentModel(o->best_model[1]);
    t.stable = o->stable; t.mateMin = o->mate_min; t.mateMax = o->mate_max;
    return t;
}

// ---- MatchSelector::determineTemplateLength (MatchSelector.cpp:188-256)
int oracle_determine_tls(oracle_ref *r, const oracle_params *cp, const uint8_t *contig_loaded, const uint8_t *bcl, uint32_t tile,
                         const uint64_t *matches, uint64_t n_matches, oracle_tls *out)
{
    try
    {
        const Params p = toParams(cp);
        const std::shared_ptr<const ContigList> contigsHeld = filteredContigsShared(r, contig_loaded); const ContigList &contigs = *contigsHeld;
        MatchSelector ms(p, contigs);
        const Match *mb = reinterpret_cast<const Match *>(matches);
        toTls(ms.determineTemplateLength(mb, mb + n_matches, bcl, tile), out);
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- MatchSelector::processMatchList for the whole tile (MatchSelector.cpp:258-368); n_threads > 1 splits the match list
// at cluster boundaries (records are concatenated in cluster order; counters[0..1] = rescue calls, rescue candidate positions)
int oracle_select(oracle_ref *r, const oracle_params *cp, const uint8_t *contig_loaded, const uint8_t *bcl, uint32_t tile,
                  const uint64_t *matches, uint64_t n_matches, const oracle_tls *tls, uint32_t n_threads,
                  void *records_out /* FragmentRecord */, uint64_t capacity, uint64_t *n_out, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar,
                  uint64_t *counters)
{
    try
    {
        const Params p = toParams(cp);
        const std::shared_ptr<const ContigList> contigsHeld = filteredContigsShared(r, contig_loaded); const ContigList &contigs = *contigsHeld;
        const TemplateLengthStatistics t = fromTls(tls);
        const Match *mb = reinterpret_cast<const Match *>(matches), *me = mb + n_matches;
        if (!n_threads) n_threads = 1;
        // pieces of the match list, cut at cluster boundaries, handed out as the threads ask for them: sixteen per thread, so that a thread that meets
        // a repeat family does not keep the others waiting for its share (a static split into one range per thread did, on 256 threads)
        const uint32_t n_pieces = 1 == n_threads ? 1 : n_threads * 16;
        std::vector<const Match *> bounds(1, mb);
        for (uint32_t i = 1; i < n_pieces; ++i)
        {
            const Match *b = mb + n_matches * i / n_pieces;
            while (b != me && b != mb && SeedId(b->seedId).getCluster() == SeedId((b - 1)->seedId).getCluster()) ++b;
            if (b < bounds.back()) b = bounds.back();
            bounds.push_back(b);
        }
        bounds.push_back(me);
        std::vector<std::vector<FragmentRecord> > recs(n_pieces);
        std::vector<std::vector<uint32_t> > cigs(n_pieces);
        std::vector<std::string> errors(n_threads);
        std::vector<uint64_t> calls(n_threads, 0), cands(n_threads, 0);
        std::atomic<uint32_t> nextPiece(0);
        std::vector<std::thread> threads;
        for (uint32_t i = 0; i < n_threads; ++i)
            threads.push_back(std::thread([&, i]()
            {
                try
                {
                    MatchSelector ms(p, contigs);
                    for (uint32_t piece = nextPiece++; piece < n_pieces; piece = nextPiece++)
                        ms.selectTile(bounds[piece], bounds[piece + 1], bcl, tile, t, recs[piece], cigs[piece]);
                    calls[i] = ms.templateBuilder.rescueCalls; cands[i] = ms.templateBuilder.rescueCandidates;
                }
                catch (const std::exception &e) { errors[i] = e.what(); if (errors[i].empty()) errors[i] = "error"; }
            }));
        for (auto &th : threads) th.join();
        uint64_t n = 0, nc = 0;
        FragmentRecord *out = reinterpret_cast<FragmentRecord *>(records_out);
        if (counters) { counters[0] = 0; counters[1] = 0; }
        for (uint32_t i = 0; i < n_threads; ++i)
        {
            if (!errors[i].empty()) throw std::runtime_error(errors[i]);
            if (counters) { counters[0] += calls[i]; counters[1] += cands[i]; }
        }
        for (uint32_t i = 0; i < n_pieces; ++i)
        {
            if (n + recs[i].size() > capacity || nc + cigs[i].size() > cigar_capacity) throw std::runtime_error("record capacity");
            for (size_t k = 0; k < recs[i].size(); ++k) { out[n] = recs[i][k]; out[n].cigarOffset += uint32_t(nc); ++n; }
            memcpy(cigar_out + nc, cigs[i].data(), cigs[i].size() * 4); nc += cigs[i].size();
        }
        *n_out = n; *n_cigar = nc;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- unit-test hooks for the golden vectors --------------------------------------------------------------------
// SeedId bit layout and overflow rule (testSeedId.cpp)
int oracle_seed_id(uint64_t tile, uint64_t barcode, uint64_t cluster, uint64_t seed, uint64_t reverse, uint64_t *value)
{
    try { *value = SeedId(tile, barcode, cluster, seed, reverse).value; return 0; }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// Runs FragmentBuilder::build on literal inputs the way testSimpleIndelAligner.cpp:146-262 / testFragmentBuilder2.cpp do:
// one contig, reads given as ASCII ('n' allowed) with quality 30 (':'-'!' in the tests' getBcl), explicit seed list and
// explicit matches (seed index, reverse, contig, seed position).
int oracle_align_literal(const oracle_params *cp, const char *contig_bases, uint32_t contig_length,
                         const uint8_t *bcl, const uint64_t *matches, uint64_t n_matches, int with_gaps,
                         oracle_candidate *out, uint64_t capacity, uint64_t *n_out, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar)
{
    try
    {
        const uint64_t offsets[2] = { 0, contig_length };
        oracle_ref *r = oracle_ref_create(contig_bases, offsets, 1);
        const int rc = oracle_build_fragments(r, cp, 0, bcl, 0, matches, n_matches, with_gaps, 0, out, capacity, n_out, cigar_out, cigar_capacity, n_cigar);
        oracle_ref_destroy(r);
        return rc;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// Test harness of testSimpleIndelAligner.cpp:146-262 restated: two ungapped candidates for the same read are built by hand
// (head at the read's own offset, tail right-aligned to the reference end), then SimpleIndelAligner::alignSimpleIndels runs.
// Scores 0:-1:-2:-1:-5, gap limit 20000 (testSimpleIndelAligner.cpp:139-158).  Reverse strand of the test reads is the plain
// reversal (no complement), qualities are the test's "irrelevantQualities" string minus 33.
int oracle_simple_indel_literal(const char *read, const char *reference, int have_seeds, const uint32_t *seed_offsets /*2*/,
                                uint32_t left_clip0, uint32_t right_clip1,
                                oracle_candidate *out /*2*/, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar)
{
    try
    {
        static const std::string irrelevantQualities("CFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDGGBGGBEGEGGBGEHDHHHGGGGGDGGGG?GGGGCFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDBCFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDB");
        const std::string readS(read), referenceS(reference);
        const long referenceOffset = long(referenceS.find_first_not_of(' '));
        const std::string referenceWithoutSpaces = referenceS.substr(referenceOffset);
        const long pos = long(readS.find_first_not_of(' '));
        const std::string readWithoutSpaces = readS.substr(pos);
        Cluster cluster; cluster.nReads = 1;
        Read &rd = cluster[0];
        rd.forwardSequence.assign(readWithoutSpaces.begin(), readWithoutSpaces.end());
        rd.forwardQuality.assign(irrelevantQualities.begin(), irrelevantQualities.end());
        rd.forwardQuality.resize(rd.forwardSequence.size());
        for (size_t i = 0; i < rd.forwardQuality.size(); ++i) rd.forwardQuality[i] -= 33;
        rd.reverseSequence = rd.forwardSequence; rd.reverseQuality = rd.forwardQuality;
        std::reverse(rd.reverseSequence.begin(), rd.reverseSequence.end()); std::reverse(rd.reverseQuality.begin(), rd.reverseQuality.end());
        std::vector<ReadMetadata> reads; { ReadMetadata rm = { rd.getLength(), 0, 0, 1 }; reads.push_back(rm); }
        std::vector<SeedMetadata> seeds;
        if (have_seeds) { SeedMetadata a = { seed_offsets[0], 32, 0, 0 }, b = { seed_offsets[1], 32, 0, 1 }; seeds.push_back(a); seeds.push_back(b); }
        else
        {
            const unsigned readLength = unsigned(std::min<long>(long(readWithoutSpaces.length()), long(referenceS.length()) - pos));
            SeedMetadata a = { 0, 32, 0, 0 }, b = { readLength - 32 - 1, 32, 0, 1 }; seeds.push_back(a); seeds.push_back(b);
        }
        FragmentMetadataList list(2);
        list[0].lowClipped = (unsigned short)left_clip0; list[1].highClipped = (unsigned short)right_clip1;
        list[0].readIndex = 0; list[0].contigId = 0; list[0].position = pos - referenceOffset; list[0].firstSeedIndex = 0;
        list[1].readIndex = 0; list[1].contigId = 0; list[1].position = long(referenceS.length()) - long(readWithoutSpaces.length()) - referenceOffset; list[1].firstSeedIndex = 1;
        const SimpleIndelAligner aligner(0, -1, -2, -1, -5, 20000);
        Contig contig; contig.index = 0; contig.name = "vasja"; contig.forward.assign(referenceWithoutSpaces.begin(), referenceWithoutSpaces.end());
        ContigList contigs(1, contig);
        const std::vector<char> &referenceV = contigs[0].forward;
        Cigar cigarBuffer; cigarBuffer.reserve(1024);
        for (size_t k = 0; k < 2; ++k)
        {
            FragmentMetadata &f = list[k];
            f.cluster = &cluster; f.cigarBuffer = &cigarBuffer; f.cigarOffset = unsigned(cigarBuffer.size()); f.observedLength = rd.getLength();
            if (0 > f.position)
            {
                const long c = std::max<long>(-f.position, f.leftClipped());
                cigarBuffer.push_back(cigarEncode(unsigned(c), SOFT_CLIP)); ++f.cigarLength; f.observedLength -= unsigned(c); f.position += c;
            }
            else if (f.leftClipped())
            {
                cigarBuffer.push_back(cigarEncode(f.leftClipped(), SOFT_CLIP)); ++f.cigarLength; f.observedLength -= f.leftClipped(); f.position += f.leftClipped();
            }
            long rightClip = 0;
            if (f.rightClipped() || (f.position + long(f.observedLength) > long(referenceV.size())))
            {
                rightClip = std::max<long>(f.rightClipped(), f.position + long(f.observedLength) - long(referenceV.size()));
                f.observedLength -= unsigned(rightClip);
            }
            cigarBuffer.push_back(cigarEncode(f.observedLength, ALIGN)); ++f.cigarLength;
            if (rightClip) { cigarBuffer.push_back(cigarEncode(unsigned(rightClip), SOFT_CLIP)); ++f.cigarLength; }
            aligner.updateFragmentCigar(reads, referenceV, f, f.position, cigarBuffer, f.cigarOffset);
        }
        if (list[1].getUnclippedPosition() < list[0].getUnclippedPosition()) std::swap(list[0], list[1]);
        aligner.alignSimpleIndels(cigarBuffer, contigs, reads, seeds, list);
        uint64_t nc = 0;
        for (size_t k = 0; k < 2; ++k)
        {
            fillCandidate(out[k], list[k], 0, uint32_t(nc));
            if (nc + list[k].cigarLength > cigar_capacity) throw std::runtime_error("cigar capacity");
            memcpy(cigar_out + nc, list[k].cigarBuffer->data() + list[k].cigarOffset, list[k].cigarLength * 4); nc += list[k].cigarLength;
        }
        *n_cigar = nc;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// Test harness of testFragmentBuilder2.cpp:146-180 restated: ungapped alignment of one read against one contig, optionally followed
// by the gapped retry with the test's own accept rule.  ELAND scores 2:-1:-15:-3:25.
int oracle_fragment_builder2_literal(const char *read, const char *reference, int reverse, int have_position, int64_t position, int gapped,
                                     oracle_candidate *out, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar, uint32_t *first_mismatch_cycle)
{
    try
    {
        static const std::string irrelevantQualities("CFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDGGBGGBEGEGGBGEHDHHHGGGGGDGGGG?GGGG");
        std::string r(read); if (reverse) std::reverse(r.begin(), r.end());
        Cluster cluster; cluster.nReads = 1;
        Read &rd = cluster[0];
        rd.forwardSequence.assign(r.begin(), r.end());
        rd.forwardQuality.assign(irrelevantQualities.begin(), irrelevantQualities.end());
        if (rd.forwardQuality.size() != rd.forwardSequence.size()) throw std::runtime_error("sequence and quality must be of equal lengths");
        for (size_t i = 0; i < rd.forwardQuality.size(); ++i) rd.forwardQuality[i] -= 33;
        rd.reverseSequence = rd.forwardSequence; rd.reverseQuality = rd.forwardQuality;
        std::reverse(rd.reverseSequence.begin(), rd.reverseSequence.end()); std::reverse(rd.reverseQuality.begin(), rd.reverseQuality.end());
        std::vector<ReadMetadata> reads; { ReadMetadata a = { 100, 0, 0, 1 }, b = { 100, 1, 100, 101 }; reads.push_back(a); reads.push_back(b); }
        FragmentMetadata f; f.reverse = reverse != 0;
        if (have_position) { f.contigId = 0; f.position = position; }
        if (f.isNoMatch()) { f.contigId = 0; f.position = 0; }
        Cigar cigarBuffer; cigarBuffer.reserve(1024);
        f.cluster = &cluster; f.cigarBuffer = &cigarBuffer;
        Contig contig; contig.index = 0; contig.name = "vasja"; contig.forward.assign(reference, reference + strlen(reference));
        const UngappedAligner ungapped(2, -1, -15, -3, 25);
        const SequencingAdapterList noAdapters;
        FragmentSequencingAdapterClipper adapterClipper(noAdapters);
        adapterClipper.checkInitStrand(f, contig);
        ungapped.alignUngapped(f, cigarBuffer, reads, adapterClipper, contig);
        if (gapped)
        {
            const GappedAligner gappedAligner(200, 2, -1, -15, -3, 25);
            FragmentMetadata tmp = f;
            const unsigned matchCount = gappedAligner.alignGapped(tmp, cigarBuffer, reads, adapterClipper, contig);
            if (matchCount + BandedSmithWaterman::WIDEST_GAP_SIZE > f.getObservedLength() && (tmp.mismatchCount <= 5) &&
                (f.mismatchCount > tmp.mismatchCount) && f.logProbability < tmp.logProbability)
                f = tmp;
        }
        fillCandidate(*out, f, 0, 0);
        if (f.cigarLength > cigar_capacity) throw std::runtime_error("cigar capacity");
        if (f.cigarLength) memcpy(cigar_out, f.cigarBuffer->data() + f.cigarOffset, f.cigarLength * 4);
        *n_cigar = f.cigarLength;
        *first_mismatch_cycle = f.mismatchCycles.empty() ? 0 : f.mismatchCycles[0];
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// TestSequencingAdapter::align (lib/alignment/cppunit/testSequencingAdapter.cpp:180-203) restated: one read (given in the direction of the reference; a
// reverse alignment reads it back to front, not complemented -- the test's Read >> operator only reverses) at position 0 of one contig, a fresh
// FragmentSequencingAdapterClipper over the given adapters, checkInitStrand, alignUngapped.  ELAND scores 2:-1:-15:-3:25.
int oracle_sequencing_adapter_literal(const char *read, const char *reference, int reverse, uint32_t n_adapters, const char *const *adapter_sequences, const uint32_t *adapter_reverse,
                                      const uint32_t *adapter_clip_length, oracle_candidate *out, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar)
{
    try
    {
        static const std::string irrelevantQualities("CFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDGGBGGBEGEGGBGEHDHHHGGGGGDGGGG?GGGGDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDGDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDG");
        std::string r(read); if (reverse) std::reverse(r.begin(), r.end());
        if (r.size() > irrelevantQualities.size()) throw std::runtime_error("read longer than the test's quality string");
        Cluster cluster; cluster.nReads = 1;
        Read &rd = cluster[0];
        rd.forwardSequence.assign(r.begin(), r.end());
        rd.forwardQuality.assign(irrelevantQualities.begin(), irrelevantQualities.begin() + r.size());
        for (size_t i = 0; i < rd.forwardQuality.size(); ++i) rd.forwardQuality[i] -= 33;
        rd.reverseSequence = rd.forwardSequence; rd.reverseQuality = rd.forwardQuality;
        std::reverse(rd.reverseSequence.begin(), rd.reverseSequence.end()); std::reverse(rd.reverseQuality.begin(), rd.reverseQuality.end());
        // (the test's ReadMetadata says 100 cycles whatever the read's length: only the mismatch cycle numbers depend on it)
        std::vector<ReadMetadata> reads; { ReadMetadata a = { 100, 0, 0, 1 }, b = { 100, 1, 100, 101 }; reads.push_back(a); reads.push_back(b); }
        SequencingAdapterList adapters;
        for (uint32_t a = 0; a < n_adapters; ++a)
        {
            SequencingAdapterMetadata m; m.sequence = adapter_sequences[a]; m.reverse = 0 != adapter_reverse[a]; m.clipLength = adapter_clip_length[a];
            adapters.push_back(SequencingAdapter(m));
        }
        FragmentMetadata f; f.reverse = reverse != 0;
        f.contigId = 0; f.position = 0;
        Cigar cigarBuffer; cigarBuffer.reserve(1024);
        f.cluster = &cluster; f.cigarBuffer = &cigarBuffer;
        Contig contig; contig.index = 0; contig.name = "vasja"; contig.forward.assign(reference, reference + strlen(reference));
        const UngappedAligner ungapped(2, -1, -15, -3, 25);
        FragmentSequencingAdapterClipper adapterClipper(adapters);
        adapterClipper.checkInitStrand(f, contig);
        ungapped.alignUngapped(f, cigarBuffer, reads, adapterClipper, contig);
        fillCandidate(*out, f, 0, 0);
        if (f.cigarLength > cigar_capacity) throw std::runtime_error("cigar capacity");
        if (f.cigarLength) memcpy(cigar_out, f.cigarBuffer->data() + f.cigarOffset, f.cigarLength * 4);
        *n_cigar = f.cigarLength;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- literal entry points for lib/alignment/cppunit/testSemialignedClipper.cpp and testOverlappingEndsClipper.cpp
static void literalRead(Read &rd, const std::string &sequence, const std::string &quality)
{
    rd.forwardSequence.assign(sequence.begin(), sequence.end());
    rd.forwardQuality.assign(quality.begin(), quality.end());
    if (rd.forwardQuality.size() != rd.forwardSequence.size()) throw std::runtime_error("sequence and quality must be of equal lengths");
    for (size_t i = 0; i < rd.forwardQuality.size(); ++i) rd.forwardQuality[i] -= 33;
    rd.reverseSequence = rd.forwardSequence; rd.reverseQuality = rd.forwardQuality;
    std::reverse(rd.reverseSequence.begin(), rd.reverseSequence.end()); std::reverse(rd.reverseQuality.begin(), rd.reverseQuality.end());
}
static Contig literalContig(const std::string &forward, long &firstPosOffset)   // makeContig of the tests: leading spaces shift the start
{
    const size_t begin = std::min(forward.find_first_not_of(' '), forward.size());
    Contig c; c.index = 0; c.name = "vasja"; c.forward.assign(forward.begin() + begin, forward.end());
    firstPosOffset = -long(begin);
    return c;
}
// TestSemialignedClipper::align (testSemialignedClipper.cpp:163-187): ungapped alignment at the position the reference string implies,
// then SemialignedEndsClipper::clip.  out: CIGAR words and the strand position
int oracle_semialigned_clip_literal(const char *read, const char *reference, int reverse, uint32_t *cigar_out, uint64_t cigar_capacity, uint64_t *n_cigar, int64_t *position_out)
{
    try
    {
        static const std::string irrelevantQualities("CFCEEBFHEHDGBDBEDDEGEHHFHEGBHHDDDB<F>FGGBFGGFGCGGGDGGDDFHHHFEGGBGDGGBGGBEGEGGBGEHDHHHGGGGGDGGGG?GGGG");
        std::string r(read); if (reverse) std::reverse(r.begin(), r.end());
        Cluster cluster; cluster.nReads = 1;
        literalRead(cluster[0], r, irrelevantQualities);
        std::vector<ReadMetadata> reads; { ReadMetadata a = { 100, 0, 0, 1 }, b = { 100, 1, 100, 101 }; reads.push_back(a); reads.push_back(b); }
        FragmentMetadata f; f.reverse = reverse != 0;
        if (f.isNoMatch()) { f.contigId = 0; f.position = 0; }
        Cigar cigarBuffer; cigarBuffer.reserve(1024);
        f.cluster = &cluster; f.cigarBuffer = &cigarBuffer;
        ContigList contigs; long offset = 0;
        contigs.push_back(literalContig(reference, offset));
        f.position = offset;
        const UngappedAligner ungapped(2, -1, -15, -3, 25);
        const SequencingAdapterList noAdapters;
        FragmentSequencingAdapterClipper adapterClipper(noAdapters);
        adapterClipper.checkInitStrand(f, contigs[0]);
        ungapped.alignUngapped(f, cigarBuffer, reads, adapterClipper, contigs[0]);
        SemialignedEndsClipper clipper; clipper.cigarBuffer.reserve(1024);
        clipper.clip(contigs, f);
        if (f.cigarLength > cigar_capacity) throw std::runtime_error("cigar capacity");
        if (f.cigarLength) memcpy(cigar_out, f.cigarBuffer->data() + f.cigarOffset, f.cigarLength * 4);
        *n_cigar = f.cigarLength; *position_out = f.position;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
// TestOverlappingEndsClipper::init + clip (testOverlappingEndsClipper.cpp:109-222): both reads placed as "xM" at the columns the
// strings imply, then OverlappingEndsClipper::clip.  out per read: CIGAR words (capacity 8), count, position
int oracle_overlapping_clip_literal(const char *read1, const char *quality1, int reverse1, const char *read2, const char *quality2, int reverse2, const char *reference,
                                    uint32_t *cigar_out /* 2 x 8 */, uint32_t *n_cigar /* 2 */, int64_t *position_out /* 2 */)
{
    try
    {
        const std::string r[2] = { read1, read2 }, q[2] = { quality1, quality2 };
        const bool rev[2] = { reverse1 != 0, reverse2 != 0 };
        ContigList contigs; long offset = 0;
        contigs.push_back(literalContig(reference, offset));
        Cluster cluster; cluster.nReads = 2;
        size_t start[2]; std::vector<ReadMetadata> reads;
        for (unsigned i = 0; i < 2; ++i)
        {
            start[i] = r[i].find_first_not_of(' ');
            std::string s = r[i].substr(start[i]); if (rev[i]) std::reverse(s.begin(), s.end());     // ReadInit reverses the string of a reverse read
            std::string ql = q[i]; if (!rev[i]) std::reverse(ql.begin(), ql.end());                  // ... and, as written there, the qualities of a forward one
            literalRead(cluster[i], s, ql);
            ReadMetadata m = { unsigned(s.size()), i, i ? reads[0].length : 0u, i ? reads[0].length + 1 : 1u }; reads.push_back(m);
        }
        Cigar cigarBuffer; cigarBuffer.reserve(1024);
        BamTemplate templ(cigarBuffer);
        templ.initialize(reads, cluster);
        for (unsigned i = 0; i < 2; ++i)
        {
            FragmentMetadata &f = templ.getFragmentMetadata(i);
            f.reverse = rev[i]; f.cigarBuffer = &cigarBuffer; f.cigarOffset = unsigned(cigarBuffer.size());
            cigarBuffer.push_back(cigarEncode(reads[i].length, ALIGN));
            f.cigarLength = unsigned(cigarBuffer.size()) - f.cigarOffset;
            f.contigId = 0; f.position = long(start[i]); f.observedLength = reads[i].length;
        }
        OverlappingEndsClipper clipper; clipper.cigarBuffer.reserve(1024);
        clipper.clip(contigs, templ);
        for (unsigned i = 0; i < 2; ++i)
        {
            const FragmentMetadata &f = templ.getFragmentMetadata(i);
            if (f.cigarLength > 8) throw std::runtime_error("cigar capacity");
            memcpy(cigar_out + 8 * i, f.cigarBuffer->data() + f.cigarOffset, f.cigarLength * 4);
            n_cigar[i] = f.cigarLength; position_out[i] = f.position;
        }
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- the fixture of lib/alignment/cppunit/testTemplateBuilder.cpp (:57-137 and BuilderInit.hh): five contigs, a cluster copied from
// contig `bcl_contig` (read 1 forward at offset0, read 2 from the reverse strand at offset1, all Q40), the dummy template length
// statistics (150/190/250, 20/30, FR+/RF-), ELAND scores and an explicit candidate list per read.
typedef struct
{
    uint32_t contig_id; int64_t position; uint32_t observed_length, read_index, reverse, cigar_offset, cigar_length, mismatch_count;
    double log_probability; uint32_t unique_seed_count, alignment_score, no_match;
} oracle_literal_fragment;
int oracle_template_builder_literal(const char *const *contig_bases, uint32_t n_contigs, uint32_t bcl_contig, int32_t offset0, int32_t offset1,
                                    const oracle_literal_fragment *frags0, uint32_t n0, const oracle_literal_fragment *frags1, uint32_t n1,
                                    uint32_t repeats /* buildTemplate calls with the same input */, uint32_t *template_score_out, oracle_literal_fragment *out /* 2 */)
{
    try
    {
        ContigList contigs;
        for (uint32_t i = 0; i < n_contigs; ++i) { Contig c; c.index = 0; c.name = "c"; c.forward.assign(contig_bases[i], contig_bases[i] + strlen(contig_bases[i])); contigs.push_back(c); }
        Params p;
        p.gapMatchScore = 2; p.gapMismatchScore = -1; p.gapOpenScore = -15; p.gapExtendScore = -3; p.minGapExtendScore = 25;     // ELAND_* of the test
        p.repeatThreshold = 10; p.gappedMismatchesMax = 8; p.semialignedGapLimit = 20000; p.scatterRepeats = false;
        p.dodgyAlignmentScore = TemplateBuilder::DODGY_ALIGNMENT_SCORE_UNALIGNED;
        { ReadMetadata a = { 100, 0, 0, 1 }, b = { 100, 1, 100, 101 }; p.reads.push_back(a); p.reads.push_back(b); }
        // getBcl (BuilderInit.hh:141-169)
        const std::vector<char> &forward = contigs.at(bcl_contig).forward;
        std::vector<char> reverse;
        for (size_t i = forward.size(); i-- > 0;) { const char b = forward[i]; reverse.push_back(b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : 'N'); }
        std::string bases(forward.begin() + offset0, forward.begin() + offset0 + 100);
        bases += std::string(reverse.begin() + offset1, reverse.begin() + offset1 + 100);
        std::vector<uint8_t> bcl;
        for (size_t i = 0; i < bases.size(); ++i) { const char b = bases[i]; bcl.push_back(uint8_t((40 << 2) | (b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3))); }
        Cluster cluster; cluster.init(p.reads, bcl.data(), 32, 1234, true);
        const RestOfGenomeCorrection rog(contigs, p.reads);
        const TemplateLengthStatistics tls(150, 250, 190, 20, 30, TemplateLengthStatistics::FRp, TemplateLengthStatistics::RFm, -1);
        const std::vector<uint32_t> cigarBuffer(1000, 1600);
        std::vector<FragmentMetadataList> fragments(2);
        const oracle_literal_fragment *in[2] = { frags0, frags1 }; const uint32_t n[2] = { n0, n1 };
        for (unsigned r = 0; r < 2; ++r) for (uint32_t i = 0; i < n[r]; ++i)
        {
            const oracle_literal_fragment &l = in[r][i];
            FragmentMetadata f;
            f.contigId = l.contig_id; f.position = l.position; f.observedLength = l.observed_length; f.readIndex = l.read_index; f.reverse = l.reverse != 0;
            f.cigarOffset = l.cigar_offset; f.cigarLength = l.cigar_length; f.cigarBuffer = &cigarBuffer; f.mismatchCount = l.mismatch_count;
            f.logProbability = l.log_probability; f.uniqueSeedCount = l.unique_seed_count; f.alignmentScore = l.alignment_score; f.cluster = &cluster;
            fragments[r].push_back(f);
        }
        TemplateBuilder builder(p);
        for (uint32_t k = 0; k < (repeats ? repeats : 1); ++k) builder.buildTemplate(contigs, rog, p.reads, fragments, cluster, tls);
        *template_score_out = builder.bamTemplate.getAlignmentScore();
        for (unsigned i = 0; i < 2; ++i)
        {
            const FragmentMetadata &f = builder.bamTemplate.getFragmentMetadata(i);
            oracle_literal_fragment &o = out[i];
            o.contig_id = f.contigId; o.position = f.position; o.observed_length = f.observedLength; o.read_index = f.readIndex; o.reverse = f.reverse;
            o.cigar_offset = f.cigarOffset; o.cigar_length = f.cigarLength; o.mismatch_count = f.mismatchCount; o.log_probability = f.logProbability;
            o.unique_seed_count = f.uniqueSeedCount; o.alignment_score = f.alignmentScore; o.no_match = f.isNoMatch();
        }
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- lib/alignment/cppunit/testShadowAligner.cpp:56-252: reads of 81 and 92 bases copied from a contig (getBcl of BuilderInit.hh with
// explicit strands), an orphan on read 1 at position 0, its mate rescued; then the mate found is used as the orphan and read 1 rescued.
// out[0] = the rescued mate, out[1] = read 1 rescued back; ok[k] = what rescueShadow returned
int oracle_shadow_aligner_literal(const char *const *contig_bases, uint32_t n_contigs, uint32_t bcl_contig, int32_t offset0, int32_t offset1, int reverse0, int reverse1,
                                  uint32_t tls_min, uint32_t tls_max, uint32_t tls_median, uint32_t tls_low, uint32_t tls_high, int model0, int model1,
                                  int orphan_reverse, oracle_literal_fragment *out /* 2 */, uint32_t *first_cigar_word /* 2 */, uint32_t *ok /* 2 */)
{
    try
    {
        ContigList contigs;
        for (uint32_t i = 0; i < n_contigs; ++i) { Contig c; c.index = 0; c.name = "c"; c.forward.assign(contig_bases[i], contig_bases[i] + strlen(contig_bases[i])); contigs.push_back(c); }
        Params p;
        p.gapMatchScore = 2; p.gapMismatchScore = -1; p.gapOpenScore = -15; p.gapExtendScore = -3; p.minGapExtendScore = 25; p.gappedMismatchesMax = 8;
        { ReadMetadata a = { 81, 0, 0, 1 }, b = { 92, 1, 81, 82 }; p.reads.push_back(a); p.reads.push_back(b); }
        const std::vector<char> &forward = contigs.at(bcl_contig).forward;
        std::vector<char> reverse;
        for (size_t i = forward.size(); i-- > 0;) { const char b = forward[i]; reverse.push_back(b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : 'N'); }
        const std::vector<char> &s0 = reverse0 ? reverse : forward, &s1 = reverse1 ? reverse : forward;
        std::string bases(s0.begin() + offset0, s0.begin() + offset0 + 81);
        bases += std::string(s1.begin() + offset1, s1.begin() + offset1 + 92);
        std::vector<uint8_t> bcl;
        for (size_t i = 0; i < bases.size(); ++i) { const char b = bases[i]; bcl.push_back(uint8_t((40 << 2) | (b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3))); }
        Cluster cluster; cluster.init(p.reads, bcl.data(), 1101, 999, true);
        const TemplateLengthStatistics tls(tls_min, tls_max, tls_median, tls_low, tls_high, TemplateLengthStatistics::AlignmentModel(model0), TemplateLengthStatistics::AlignmentModel(model1), -1);
        ShadowAligner shadowAligner(p);
        FragmentMetadataList shadowList; shadowList.reserve(50);
        FragmentMetadata orphan; orphan.cluster = &cluster; orphan.readIndex = 0; orphan.contigId = bcl_contig; orphan.position = 0; orphan.reverse = orphan_reverse != 0;
        for (unsigned k = 0; k < 2; ++k)
        {
            shadowList.clear();
            ok[k] = shadowAligner.rescueShadow(contigs, orphan, shadowList, 50, p.reads, tls, 0);
            if (!ok[k] || shadowList.empty()) { ok[k] = 0; break; }
            const FragmentMetadata f = shadowList[0];
            oracle_literal_fragment &o = out[k];
            o.contig_id = f.contigId; o.position = f.position; o.observed_length = f.observedLength; o.read_index = f.readIndex; o.reverse = f.reverse;
            o.cigar_offset = f.cigarOffset; o.cigar_length = f.cigarLength; o.mismatch_count = f.mismatchCount; o.log_probability = f.logProbability;
            o.unique_seed_count = f.uniqueSeedCount; o.alignment_score = f.alignmentScore; o.no_match = f.isNoMatch();
            // the test reads getCigarBuffer()[0]: on its own genome the mate is the first candidate of the window.  With other draws of
            // the contigs chance 7-mer hits come first, so the word reported is the rescued fragment's own first CIGAR word
            first_cigar_word[k] = f.cigarLength ? shadowAligner.shadowCigarBuffer.at(f.cigarOffset) : 0;
            orphan = f;
        }
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- literal entry points for the known-answer vectors of lib/alignment/cppunit/testTemplateLengthStatistics.cpp
int oracle_tls_alignment_model(int64_t pos1, int reverse1, int64_t pos2, int reverse2)
{
    FragmentMetadata f1, f2; f1.position = pos1; f1.reverse = reverse1 != 0; f2.position = pos2; f2.reverse = reverse2 != 0;
    return int(TemplateLengthStatistics::alignmentModel(f1, f2));
}
int oracle_tls_alignment_class(int model) { return int(TemplateLengthStatistics::alignmentClass(TemplateLengthStatistics::AlignmentModel(model))); }
// out: { mateOrientation, mateMinPosition, mateMaxPosition }
void oracle_tls_mate(uint32_t mn, uint32_t mx, uint32_t median, uint32_t low, uint32_t high, int model0, int model1, int drift,
                     uint32_t read_index, int reverse, int64_t position, uint32_t len0, uint32_t len1, int64_t *out)
{
    const TemplateLengthStatistics tls(mn, mx, median, low, high, TemplateLengthStatistics::AlignmentModel(model0), TemplateLengthStatistics::AlignmentModel(model1), drift);
    const unsigned readLengths[2] = { len0, len1 };
    out[0] = tls.mateOrientation(read_index, reverse != 0);
    out[1] = tls.mateMinPosition(read_index, reverse != 0, position, readLengths);
    out[2] = tls.mateMaxPosition(read_index, reverse != 0, position, readLengths);
}
// TestTemplateLengthStatistics::addTemplates (:98-128): the literal sequence of templates.  out: after the first 10000 templates
// { min, median, max, lowStdDev, highStdDev, mateMin, mateMax }, then { value returned by the last addTemplate, stable }
void oracle_tls_add_templates_sequence(int drift, uint32_t *out)
{
    TemplateLengthDistribution tls(drift);
    const std::vector<uint32_t> cigarBuffer(1, 16);
    std::vector<FragmentMetadataList> f(2, FragmentMetadataList(1));
    f[0][0].contigId = 0; f[0][0].position = 0; f[0][0].observedLength = 1; f[0][0].reverse = false;
    f[0][0].cigarBuffer = &cigarBuffer; f[0][0].cigarOffset = 0; f[0][0].cigarLength = 1;
    f[1][0] = f[0][0]; f[1][0].reverse = true;
    bool any = false;
    for (unsigned i = 1; i < 10000; ++i) { any |= tls.addTemplate(f); ++f[1][0].position; }
    std::swap(f[0], f[1]);
    any |= tls.addTemplate(f);
    out[0] = tls.stats.min; out[1] = tls.stats.median; out[2] = tls.stats.max; out[3] = tls.stats.lowStdDev; out[4] = tls.stats.highStdDev;
    out[5] = tls.stats.mateMin; out[6] = tls.stats.mateMax; out[7] = any;
    std::swap(f[0], f[1]);
    f[1][0].position = f[0][0].position;
    for (unsigned i = 1; i < 10000; ++i) { any |= tls.addTemplate(f); ++f[1][0].position; }
    out[8] = tls.addTemplate(f); out[9] = tls.isStable();
}


// ---- bit layouts and oligo helpers behind the seed lookup and the index builder, as the reference's small unit tests drive them
// matchFinder::ClusterInfo (testMatchFinderClusterInfo.cpp:38-91).  ops: 0 = markReadComplete(arg), 1 = setBarcodeIndex(arg),
// 2 = unmarkComplete; state_out: { getBarcodeIndex, isBarcodeSet, isReadComplete(0), isReadComplete(1), byte1, byte2 }
int oracle_cluster_info(int mark_complete_ctor /* -1: default constructor */, const uint32_t *ops, const uint32_t *args, uint32_t n_ops, uint32_t *state_out)
{
    try
    {
        matchFinder::ClusterInfo c = mark_complete_ctor < 0 ? matchFinder::ClusterInfo() : matchFinder::ClusterInfo(mark_complete_ctor != 0);
        for (uint32_t i = 0; i < n_ops; ++i)
        {
            if (0 == ops[i]) c.markReadComplete(args[i]);
            else if (1 == ops[i]) c.setBarcodeIndex(args[i]);
            else c.unmarkComplete();
        }
        state_out[0] = c.getBarcodeIndex(); state_out[1] = c.isBarcodeSet(); state_out[2] = c.isReadComplete(0); state_out[3] = c.isReadComplete(1);
        state_out[4] = c.byte(0); state_out[5] = c.byte(1);
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// oligo::KmerGenerator<unsigned>::next until exhausted (testKmerGenerator.cpp:37-77)
int oracle_kmer_generator(const char *sequence, uint64_t length, uint32_t kmer_length, uint32_t *kmers_out, int64_t *positions_out, uint64_t capacity, uint64_t *n_out)
{
    KmerGenerator<unsigned> g(sequence, sequence + length, kmer_length);
    unsigned kmer; const char *position; uint64_t n = 0;
    while (g.next(kmer, position))
    {
        if (n >= capacity) { g_error = "k-mer capacity"; return 1; }
        kmers_out[n] = kmer; positions_out[n] = position - sequence; ++n;
    }
    *n_out = n;
    return 0;
}
// oligo::generateKmer / getMaxKmer<unsigned long> (testKmerGenerator.cpp:80-97)
int oracle_generate_kmer(uint32_t kmer_length, const char *sequence, uint64_t length, uint32_t *kmer_out)
{ unsigned kmer = 0; const bool ok = generateKmer<unsigned>(kmer_length, kmer, sequence, sequence + length); *kmer_out = kmer; return ok; }
uint64_t oracle_max_kmer(uint32_t kmer_length) { return getMaxKmer<uint64_t>(kmer_length); }

} // extern "C"
typedef unsigned __int128 u128;
static u128 make128(uint64_t hi, uint64_t lo) { return (u128(hi) << 64) | lo; }
template <typename F> static void withKmerType(uint32_t kmer_bases, uint64_t hi, uint64_t lo, uint64_t *out_hi, uint64_t *out_lo, const F &f)
{
    if (16 == kmer_bases) { *out_hi = 0; *out_lo = f(uint32_t(lo)); }
    else if (32 == kmer_bases) { *out_hi = 0; *out_lo = f(uint64_t(lo)); }
    else { const u128 r = f(make128(hi, lo)); *out_hi = uint64_t(r >> 64); *out_lo = uint64_t(r); }
}
struct ApplyPermutate { const Permutate &p; bool reorder; template <typename K> K operator()(K k) const { return reorder ? p.reorder(k) : p(k); } };
extern "C" {
// oligo::Permutate(blockLength, from, to)(kmer) or .reorder(kmer) (testPermutate.cpp:41-103); k-mers of 16, 32 or 64 bases as (hi, lo)
int oracle_permutate(uint32_t block_length, const uint32_t *from, const uint32_t *to, uint32_t count, uint32_t kmer_bases, int reorder, uint64_t hi, uint64_t lo, uint64_t *out_hi, uint64_t *out_lo)
{
    try
    {
        const Permutate p(block_length, std::vector<unsigned>(from, from + count), std::vector<unsigned>(to, to + count));
        const ApplyPermutate f = { p, reorder != 0 };
        withKmerType(kmer_bases, hi, lo, out_hi, out_lo, f);
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
} // extern "C"
// the walk of testPermutate.cpp:105-118 through getPermutateList<KmerT>(errorCount): the k-mer after every permutation applied in
// turn; ok_out = every intermediate value reorders to the original, and the first permutation is the identity both ways
struct WalkPermutateList
{
    const std::vector<Permutate> &list; mutable bool ok;
    template <typename K> K operator()(K original) const
    {
        ok = original == list.front()(original) && original == list.front().reorder(original);
        K permuted = original;
        for (size_t i = 0; i < list.size(); ++i) { permuted = list[i](permuted); ok = ok && original == list[i].reorder(permuted); }
        ok = ok && original == list.back().reorder(permuted);
        return permuted;
    }
};
extern "C" {
int oracle_permutate_list_walk(uint32_t kmer_bases, uint32_t error_count, uint64_t hi, uint64_t lo, uint64_t *list_size, uint64_t *out_hi, uint64_t *out_lo, int *ok_out)
{
    try
    {
        const std::vector<Permutate> list = getPermutateList(kmer_bases, error_count);
        const WalkPermutateList w = { list, false };
        withKmerType(kmer_bases, hi, lo, out_hi, out_lo, w);
        *list_size = list.size(); *ok_out = w.ok;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
// reference::NeighborsFinder<KmerT>::findNeighbors(kmerList, jobs) on the list as given (testNeighborsFinder.cpp:44-97); 32- or 64-base k-mers
int oracle_find_neighbors(uint32_t kmer_bases, const uint64_t *hi, const uint64_t *lo, uint64_t n, uint32_t jobs, uint8_t *has_neighbors_out)
{
    if (32 == kmer_bases)
    {
        std::vector<AnnotatedKmer<uint64_t> > l(n);
        for (uint64_t i = 0; i < n; ++i) { l[i].value = lo[i]; l[i].hasNeighbors = false; }
        findNeighbors(l, jobs);
        for (uint64_t i = 0; i < n; ++i) has_neighbors_out[i] = l[i].hasNeighbors;
    }
    else if (64 == kmer_bases)
    {
        std::vector<AnnotatedKmer<u128> > l(n);
        for (uint64_t i = 0; i < n; ++i) { l[i].value = make128(hi[i], lo[i]); l[i].hasNeighbors = false; }
        findNeighbors(l, jobs);
        for (uint64_t i = 0; i < n; ++i) has_neighbors_out[i] = l[i].hasNeighbors;
    }
    else { g_error = "k-mers of 32 or 64 bases"; return 1; }
    return 0;
}

} // extern "C"

// ---- gap realigner (realign.cpp): one case of lib/build/cppunit/testGapRealigner.cpp (realign() at :353-424 of the test)
extern "C" int oracle_realign_case(const char *contig, uint64_t contig_length, const uint8_t *read_bcl, uint32_t read_length, uint64_t f_strand_position, const uint32_t *cigar, uint32_t cigar_length,
                                   uint32_t observed_length, uint32_t edit_distance, uint32_t low_clipped, uint32_t high_clipped, const int64_t *gap_positions, const int32_t *gap_lengths, uint32_t n_gaps,
                                   uint32_t mismatch_cost, uint32_t gap_open_cost, uint32_t gap_extend_cost, int vigorous, int dodgy, uint32_t gaps_per_fragment, int clip_semialigned,
                                   uint64_t bin_start, int64_t bin_end /* < 0: bin_start + contig length */,
                                   uint64_t *realigned_position, uint32_t *realigned_cigar, uint32_t *realigned_cigar_length, uint32_t *realigned_edit_distance, uint32_t *realigned_observed_length,
                                   uint32_t *overlaps, uint32_t *n_overlaps)
{
    try
    {
        ContigList contigs(1);
        contigs[0].index = 0; contigs[0].name = "testContig"; contigs[0].forward.assign(contig, contig + contig_length);
        const ReferencePosition binStartPos(0, bin_start);
        const ReferencePosition binEndPos = bin_end < 0 ? ReferencePosition(0, bin_start + contig_length) : ReferencePosition(0, uint64_t(bin_end));
        RealignerGaps realignerGaps;
        for (uint32_t i = 0; i < n_gaps; ++i) realignerGaps.gapGroups.push_back(RealignGap(ReferencePosition(0, uint64_t(gap_positions[i])), gap_lengths[i]));
        realignerGaps.finalizeGaps();
        std::vector<RealignGap> all;
        realignerGaps.findGaps(binStartPos, binEndPos, all, 100000);
        const OverlappingGapsFilter filter(all);
        *n_overlaps = uint32_t(filter.overlappingGaps.size());
        for (size_t i = 0; i < filter.overlappingGaps.size() && i < 32; ++i) overlaps[i] = filter.overlappingGaps[i];
        RealignFragment fragment = RealignFragment();
        fragment.fStrandPosition = ReferencePosition(0, f_strand_position); fragment.mateFStrandPosition = fragment.fStrandPosition.value; fragment.observedLength = observed_length;
        fragment.lowClipped = uint16_t(low_clipped); fragment.highClipped = uint16_t(high_clipped); fragment.alignmentScore = 1; fragment.templateAlignmentScore = 0;
        fragment.readLength = uint16_t(read_length); fragment.editDistance = uint16_t(edit_distance); fragment.flags = 0; fragment.bases = read_bcl;
        RealignIndex index = { fragment.fStrandPosition, cigar, cigar + cigar_length };
        const GapRealigner realigner = { vigorous != 0, dodgy != 0, gaps_per_fragment, mismatch_cost, gap_open_cost, gap_extend_cost, clip_semialigned != 0, contigs };
        std::vector<uint32_t> realignedCigars; realignedCigars.reserve(1 << 20);
        bool changed = false;
        realigner.realign(realignerGaps, binStartPos, binEndPos, index, fragment, realignedCigars, changed);
        *realigned_position = index.pos.getPosition();
        if (index.pos != fragment.fStrandPosition) throw std::logic_error("index.pos_ and fragment.fStrandPosition_ differ after realignment");
        *realigned_cigar_length = uint32_t(index.cigarEnd - index.cigarBegin);
        for (const uint32_t *it = index.cigarBegin; it != index.cigarEnd; ++it) *realigned_cigar++ = *it;
        *realigned_edit_distance = fragment.editDistance; *realigned_observed_length = fragment.observedLength;
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// ---- BAM records and header (bam.cpp) ---------------------------------------------------------------------------
extern "C" {
typedef struct { const uint8_t *bcl; const void *records; const uint32_t *cigars; uint64_t n_records; const char *read_name_prefix, *read_group; const oracle_tls *tls; } oracle_bam_tile;

// the literal index entries of lib/build/cppunit/testDuplicateFiltering.cpp through the filter: is_duplicate_out[i] for entry i as given
int oracle_filter_duplicates(uint64_t n, const uint64_t *primary, const uint64_t *mate_anchor, const uint32_t *mate_info, const uint64_t *rank, const uint64_t *cluster_id,
                             uint8_t *is_duplicate_out)
{
    try
    {
        std::vector<PairEndIndex> ends(n);
        for (uint64_t i = 0; i < n; ++i) { PairEndIndex e = { primary[i], mate_anchor[i], mate_info[i], 0, rank[i], cluster_id[i], i }; ends[i] = e; }
        std::vector<char> dup;
        filterDuplicates(ends, dup);
        for (uint64_t i = 0; i < n; ++i) is_duplicate_out[ends[i].tag] = uint8_t(dup[i]);
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// bin_cuts: ascending ReferencePosition values at which a contig goes on into a further bin (BamOptions::binCuts), or NULL
int oracle_bam_records_cuts(const oracle_bam_tile *tiles, uint32_t n_tiles, uint32_t n_reads, const uint32_t *read_lengths, uint32_t forced_dodgy_alignment_score,
                            int pessimistic_mapq, const char *read_group, const char *barcode, int mark_duplicates, int keep_duplicates,
                            int realign_gaps, int realign_dodgy, int clip_semialigned, const oracle_ref *reference, const oracle_tls *tls,
                            const uint64_t *bin_cuts, uint32_t n_cuts,
                            uint8_t *out, uint64_t capacity, uint64_t *n_bytes, uint64_t *n_records, uint64_t *unaligned_offset);
int oracle_bam_records(const oracle_bam_tile *tiles, uint32_t n_tiles, uint32_t n_reads, const uint32_t *read_lengths, uint32_t forced_dodgy_alignment_score,
                       int pessimistic_mapq, const char *read_group, const char *barcode, int mark_duplicates, int keep_duplicates,
                       int realign_gaps, int realign_dodgy, int clip_semialigned, const oracle_ref *reference, const oracle_tls *tls,
                       uint8_t *out, uint64_t capacity, uint64_t *n_bytes, uint64_t *n_records, uint64_t *unaligned_offset)
{
    return oracle_bam_records_cuts(tiles, n_tiles, n_reads, read_lengths, forced_dodgy_alignment_score, pessimistic_mapq, read_group, barcode, mark_duplicates, keep_duplicates,
                                   realign_gaps, realign_dodgy, clip_semialigned, reference, tls, 0, 0, out, capacity, n_bytes, n_records, unaligned_offset);
}
int oracle_bam_records_cuts(const oracle_bam_tile *tiles, uint32_t n_tiles, uint32_t n_reads, const uint32_t *read_lengths, uint32_t forced_dodgy_alignment_score,
                            int pessimistic_mapq, const char *read_group, const char *barcode, int mark_duplicates, int keep_duplicates,
                            int realign_gaps, int realign_dodgy, int clip_semialigned, const oracle_ref *reference, const oracle_tls *tls,
                            const uint64_t *bin_cuts, uint32_t n_cuts,
                            uint8_t *out, uint64_t capacity, uint64_t *n_bytes, uint64_t *n_records, uint64_t *unaligned_offset)
{
    try
    {
        std::vector<BamTileInput> in;
        std::vector<TemplateLengthStatistics> tileStats(n_tiles);
        for (uint32_t i = 0; i < n_tiles; ++i)
        {
            BamTileInput t = { tiles[i].bcl, static_cast<const FragmentRecord *>(tiles[i].records), tiles[i].cigars, tiles[i].n_records, tiles[i].read_name_prefix, tiles[i].read_group ? tiles[i].read_group : "" };
            if (tiles[i].tls) { tileStats[i] = fromTls(tiles[i].tls); t.tls = &tileStats[i]; }
            in.push_back(t);
        }
        BamOptions o; o.clusterLength = 0; o.readOffset[0] = o.readOffset[1] = 0;
        for (uint32_t r = 0; r < n_reads; ++r) { o.readOffset[r] = o.clusterLength; o.clusterLength += read_lengths[r]; }
        o.forcedDodgyAlignmentScore = (unsigned char)forced_dodgy_alignment_score; o.pessimisticMapQ = pessimistic_mapq; o.readGroup = read_group; o.barcode = barcode; o.markDuplicates = mark_duplicates != 0; o.keepDuplicates = keep_duplicates != 0;
        TemplateLengthStatistics stats; if (tls) stats = fromTls(tls);
        o.realignGaps = realign_gaps != 0; o.realignVigorously = 2 == realign_gaps /* --realign-vigorously 1 */; o.realignDodgy = realign_dodgy != 0; o.clipSemialigned = clip_semialigned != 0; o.contigs = reference ? &reference->contigs : 0; o.tls = tls ? &stats : 0;
        if (o.realignGaps && !o.contigs) throw std::runtime_error("gap realignment needs the reference");
        for (uint32_t k = 0; k < n_cuts; ++k) o.binCuts.push_back(bin_cuts[k] & ~uint64_t(1));
        if (!std::is_sorted(o.binCuts.begin(), o.binCuts.end())) throw std::runtime_error("bin cuts must ascend");
        std::vector<char> os;
        bamRecords(in, o, os, *n_records, *unaligned_offset);
        *n_bytes = os.size();
        if (os.size() > capacity) throw std::runtime_error("bam capacity");
        memcpy(out, os.data(), os.size());
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

// records: the uncompressed record stream; part k = records [part_offsets[k], + part_bytes[k]) compressed to bgzf[k] (bgzf_bytes[k] bytes)
int oracle_bam_index(const uint8_t *records, const uint64_t *part_offsets, const uint64_t *part_bytes, const uint8_t *const *bgzf, const uint64_t *bgzf_bytes, uint32_t n_parts,
                     uint32_t n_contigs, uint32_t header_compressed_length, uint8_t *out, uint64_t capacity, uint64_t *n_bytes)
{
    try
    {
        std::vector<BamIndexPartInput> parts(n_parts);
        for (uint32_t k = 0; k < n_parts; ++k)
        {
            parts[k].bgzf.assign(reinterpret_cast<const char *>(bgzf[k]), reinterpret_cast<const char *>(bgzf[k]) + bgzf_bytes[k]);
            const uint8_t *b = records + part_offsets[k], *end = b + part_bytes[k];
            while (b < end)
            {
                // the fields FragmentAccessorBamAdapter hands to the indexer, read back from the serialised record (Bam.hh:257-345)
                int32_t blockSize, refId, pos, lSeq; uint32_t binMqNl, flagNc;
                memcpy(&blockSize, b, 4); memcpy(&refId, b + 4, 4); memcpy(&pos, b + 8, 4); memcpy(&binMqNl, b + 12, 4); memcpy(&flagNc, b + 16, 4); memcpy(&lSeq, b + 20, 4);
                const uint32_t nameLength = binMqNl & 0xff, nCigar = flagNc & 0xffff;
                uint32_t observed = 0;
                for (uint32_t c = 0; c < nCigar; ++c)
                {
                    uint32_t w; memcpy(&w, b + 36 + nameLength + 4 * c, 4);
                    const uint32_t op = w & 15;
                    if (0 == op || 2 == op || 3 == op || 7 == op || 8 == op) observed += w >> 4;      // M D N = X: the reference bases covered
                }
                const BamIndexRecord r = { refId, pos, lSeq, observed, flagNc >> 16, uint32_t(blockSize) + 4 };
                parts[k].records.push_back(r);
                b += uint32_t(blockSize) + 4;
            }
        }
        std::vector<char> bai;
        bamIndex(parts, n_contigs, header_compressed_length, bai);
        *n_bytes = bai.size();
        if (bai.size() > capacity) throw std::runtime_error("bai capacity");
        memcpy(out, bai.data(), bai.size());
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}

int oracle_bam_header(const char *command_line, const char *description, const char *version, const char *const *header_lines, uint32_t n_header_lines,
                      const char *const *contig_names, const uint32_t *contig_lengths, const char *const *contig_as, const char *const *contig_ur, const char *const *contig_m5,
                      uint32_t n_contigs, uint8_t *out, uint64_t capacity, uint64_t *n_bytes)
{
    try
    {
        std::vector<std::string> lines(header_lines, header_lines + n_header_lines);
        std::vector<std::pair<std::string, uint32_t> > refs;
        std::vector<SqTags> tags(n_contigs);
        for (uint32_t i = 0; i < n_contigs; ++i)
        {
            refs.push_back(std::make_pair(std::string(contig_names[i]), contig_lengths[i]));
            if (contig_as && contig_as[i]) tags[i].as = contig_as[i];
            if (contig_ur && contig_ur[i]) tags[i].ur = contig_ur[i];
            if (contig_m5 && contig_m5[i]) tags[i].m5 = contig_m5[i];
        }
        std::vector<char> os;
        bamHeader(command_line, description, version, lines, refs, os, &tags);
        *n_bytes = os.size();
        if (os.size() > capacity) throw std::runtime_error("bam capacity");
        memcpy(out, os.data(), os.size());
        return 0;
    }
    catch (const std::exception &e) { g_error = e.what(); return 1; }
}
} // extern "C"

extern "C" {
// FastqSeedSource's tile rule (fastq.cpp)
int oracle_fastq_tiles(uint32_t clusters_loaded, uint32_t clusters_at_a_time, uint32_t n_seeds, uint32_t first_tile, uint32_t *numbers, uint32_t *sizes, uint32_t capacity,
                       uint32_t *n_tiles, uint32_t *next_tile)
{
    std::vector<std::pair<unsigned, unsigned> > tiles;
    unsigned current = first_tile;
    fastqDiscoverTiles(clusters_loaded, fastqTileClustersMax(clusters_at_a_time, n_seeds), current, tiles);
    *n_tiles = uint32_t(tiles.size()); *next_tile = current;
    for (size_t i = 0; i < tiles.size() && i < capacity; ++i) { numbers[i] = tiles[i].first; sizes[i] = tiles[i].second; }
    return 0;
}
} // extern "C"

