// A host written against include/isaac_gpu.h alone (no Python, no torch): what the reference-side binding of INTEGRATION.md does for
// one tile -- reference::loadContigs -> isaac-sort-reference -> FindMatchesTransition -> MatchSelector::determineTemplateLength ->
// MatchSelector::processMatchList -- on a synthetic contig and synthetic pairs, checked against the simulation's truth.
// Build: hipcc -std=c++17 -I include tests/host_example/align_tile.cpp -L isaac_aligner_amd -lisaac_gpu -o align_tile
#include "isaac_gpu.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(call) do { const int rc_ = (call); if (rc_) { std::fprintf(stderr, "%s: error %d: %s\n", #call, rc_, isaac_gpu_last_error()); return 1; } } while (0)

static uint64_t rngState = 88172645463325252ull;
static uint32_t rnd() { rngState ^= rngState << 13; rngState ^= rngState >> 7; rngState ^= rngState << 17; return uint32_t(rngState >> 11); }

int main(int argc, char **argv)
{
    const uint32_t nClusters = argc > 1 ? uint32_t(std::atoi(argv[1])) : 20000;
    const uint64_t contigLength = 2000000;
    const uint32_t L = 150;
    // --- the tile's inputs
    std::string contig(contigLength, 'A');
    for (auto &c : contig) c = "ACGT"[rnd() & 3];
    std::vector<uint8_t> bcl(size_t(nClusters) * 2 * L);
    std::vector<uint64_t> truth(nClusters);
    auto code = [](char c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; };
    for (uint32_t k = 0; k < nClusters; ++k)
    {
        const uint32_t insert = 300 + rnd() % 100;
        const uint64_t start = rnd() % (contigLength - insert);
        const bool flip = rnd() & 1;                    // the fragment comes from the reverse strand: read 1 is its far end
        truth[k] = flip ? start + insert - L : start;   // leftmost reference base of read 1
        for (uint32_t i = 0; i < L; ++i)
        {
            // the read that starts the fragment is forward, its mate the reverse complement of the fragment's end
            uint32_t fwd = code(contig[start + i]), rev = 3u - code(contig[start + insert - 1 - i]);
            if (rnd() % 300 == 0) fwd = (fwd + 1 + rnd() % 3) & 3;
            if (rnd() % 300 == 0) rev = (rev + 1 + rnd() % 3) & 3;
            const uint32_t q = 30 + rnd() % 10;
            bcl[size_t(k) * 2 * L + i] = uint8_t((flip ? rev : fwd) | (q << 2));
            bcl[size_t(k) * 2 * L + L + i] = uint8_t((flip ? fwd : rev) | (q << 2));
        }
    }
    // --- options::AlignOptions defaults for 2 x 150 (--seeds auto: 0, 118, 32, 64 per read; first two of each read in pass 0)
    isaac_params p; std::memset(&p, 0, sizeof(p));
    p.gap_match = 0; p.gap_mismatch = -3; p.gap_open = -11; p.gap_extend = -4; p.min_gap_extend = -20;
    p.repeat_threshold = 10; p.gapped_mismatches_max = 5; p.semialigned_gap_limit = 100; p.base_quality_cutoff = 25;
    p.clip_semialigned = 1; p.clip_overlapping = 1; p.dodgy_alignment_score = 0; p.keep_unaligned = 1; p.mate_drift_range = -1;
    p.first_pass_seeds = 2; p.seed_length = 32; p.n_reads = 2; p.read_length[0] = p.read_length[1] = L;
    const uint16_t offsets[4] = { 0, 118, 32, 64 };
    for (uint32_t r = 0; r < 2; ++r) for (uint32_t s = 0; s < 4; ++s) { isaac_seed &seed = p.seeds[p.n_seeds++]; seed.offset = offsets[s]; seed.length = 32; seed.read_index = r; }

    isaac_gpu_ctx *ctx = nullptr;
    CHECK(isaac_gpu_create(0, &p, nullptr, &ctx));
    const uint64_t contigOffsets[2] = { 0, contigLength };
    CHECK(isaac_gpu_load_contigs(ctx, contig.data(), contigOffsets, 1));
    uint64_t nEntries = 0;
    CHECK(isaac_gpu_build_index(ctx, 1000, 1, &nEntries));
    void *bclDev = nullptr, *matchesDev = nullptr, *offsetsDev = nullptr, *recordsDev = nullptr, *cigarDev = nullptr;
    const uint64_t matchCapacity = uint64_t(nClusters) * 72;
    CHECK(isaac_gpu_malloc(ctx, bcl.size(), &bclDev));
    CHECK(isaac_gpu_malloc(ctx, matchCapacity * sizeof(isaac_match), &matchesDev));
    CHECK(isaac_gpu_malloc(ctx, (uint64_t(nClusters) + 1) * 8, &offsetsDev));
    CHECK(isaac_gpu_malloc(ctx, uint64_t(nClusters) * 2 * sizeof(isaac_fragment), &recordsDev));
    CHECK(isaac_gpu_malloc(ctx, uint64_t(nClusters) * 2 * ISAAC_GPU_MAX_CIGAR_OPS * 4, &cigarDev));
    CHECK(isaac_gpu_upload(ctx, bclDev, bcl.data(), bcl.size()));
    uint64_t nMatches = 0; uint8_t contigHasMatches[1] = { 0 };
    CHECK(isaac_gpu_find_matches(ctx, static_cast<const uint8_t *>(bclDev), nClusters, 1, static_cast<isaac_match *>(matchesDev), matchCapacity, static_cast<uint64_t *>(offsetsDev), &nMatches,
                                 contigHasMatches));
    CHECK(isaac_gpu_set_loaded_contigs(ctx, contigHasMatches, 1));
    isaac_tls tls;
    CHECK(isaac_gpu_determine_tls(ctx, static_cast<const uint8_t *>(bclDev), nClusters, 1, static_cast<const isaac_match *>(matchesDev), static_cast<const uint64_t *>(offsetsDev), &tls));
    CHECK(isaac_gpu_select(ctx, static_cast<const uint8_t *>(bclDev), nClusters, 1, static_cast<const isaac_match *>(matchesDev), static_cast<const uint64_t *>(offsetsDev), &tls,
                           static_cast<isaac_fragment *>(recordsDev), static_cast<uint32_t *>(cigarDev), uint64_t(nClusters) * 2 * ISAAC_GPU_MAX_CIGAR_OPS));
    std::vector<isaac_fragment> records(size_t(nClusters) * 2);
    CHECK(isaac_gpu_download(ctx, records.data(), recordsDev, records.size() * sizeof(isaac_fragment)));
    isaac_counters counters;
    CHECK(isaac_gpu_get_counters(ctx, &counters));
    // --- the first read of every pair must come back where it was taken from
    uint32_t placed = 0, confident = 0, proper = 0;
    for (uint32_t k = 0; k < nClusters; ++k)
    {
        const isaac_fragment &f = records[size_t(k) * 2];
        const uint64_t position = (f.f_strand_position >> 1) & ((uint64_t(1) << 40) - 1);
        const bool aligned = !(f.flags & 2);
        if (aligned && f.mapq >= 30) { ++confident; if (position + 10 >= truth[k] && position <= truth[k] + 10) ++placed; }
        if (f.flags & 256) ++proper;
    }
    std::printf("index entries %llu, matches %llu, tls median %u, confident first reads %u of %u, placed at the truth %u, proper pairs %u, rescue calls %llu\n",
                (unsigned long long)nEntries, (unsigned long long)nMatches, tls.median, confident, nClusters, placed, proper, (unsigned long long)counters.rescue_calls);
    const bool ok = nEntries > contigLength / 2 && confident > nClusters * 9 / 10 && placed >= confident - confident / 200 && proper > nClusters * 9 / 10 && tls.median > 300 && tls.median < 400 &&
                    counters.rescue_calls > nClusters / 4;
    isaac_gpu_free(ctx, bclDev); isaac_gpu_free(ctx, matchesDev); isaac_gpu_free(ctx, offsetsDev); isaac_gpu_free(ctx, recordsDev); isaac_gpu_free(ctx, cigarDev);
    isaac_gpu_destroy(ctx);
    std::printf(ok ? "host example: ok\n" : "host example: FAILED\n");
    return ok ? 0 : 1;
}
