// CPU debugging harness for the thread-serial device logic of isaac_aligner_amd/csrc/*.h.
//
// TEST INFRASTRUCTURE ONLY.  The container that builds this repository has no GPU; this file compiles the very headers the
// gfx950 kernels are made of with g++ and runs the per-cluster functions in a plain loop, so that their logic can be checked
// against the oracle before a GPU box is spent on them.  It is not part of the product library and the product has no CPU path.
#include "../../isaac_aligner_amd/csrc/cluster_ops.h"
#include "../../isaac_aligner_amd/csrc/template_lean.h"
#include "../../isaac_aligner_amd/csrc/fragment_lean.h"
#include "../../isaac_aligner_amd/csrc/sums.h"
#include "../../isaac_aligner_amd/csrc/bam_kernels.h"
#include "../../isaac_aligner_amd/csrc/bgzf_kernels.h"
#include "../../isaac_aligner_amd/csrc/deflate_common.h"
#include "../../isaac_aligner_amd/csrc/realign.h"
#include "../../isaac_aligner_amd/csrc/host_util.h"
#include <string>
#include <vector>
#include <chrono>
#include <cstring>
#include <algorithm>

using namespace isaac;

namespace {
thread_local std::string g_error;

struct Emu
{
    DevParams P; DevReference R;
    std::vector<char> bases; std::vector<u64> offsets; std::vector<u8> loaded;
    std::vector<double> logMatch, logMismatch;
    DevAdapters adapters; std::vector<u32> adapterRanges;        // --default-adapters: the list, and four ranges per cluster (k_adapter_ranges)
    std::vector<ClusterStore> stores;       // fixed-capacity backing of the views below (the device keeps compact pools instead)
    std::vector<ClusterFragments> frags;
    std::vector<Match> matches; std::vector<u64> matchOffsets;
    Counters cnt; bool flatRescue = true; bool lean = true; double *clusterTimes = nullptr; bool fastSort = true; u32 sumsCap = 0; int sumsRadixMin = -1; std::vector<u32> dbgJobBase; std::vector<RescueJob> dbgJobs;
};
}

extern "C" {

const char *emu_last_error() { return g_error.c_str(); }

Emu *emu_create(const isaac_params *p, const char *bases, const u64 *offsets, u32 nContigs, const u8 *loaded)
{
    try
    {
        Emu *e = new Emu;
        e->P = makeDevParams(*p);
        e->adapters = makeDevAdapters(*p);
        if (e->adapters.n) e->P.adapters = &e->adapters;
        e->bases.assign(bases, bases + offsets[nContigs]); e->offsets.assign(offsets, offsets + nContigs + 1);
        e->loaded.assign(nContigs, 1); if (loaded) e->loaded.assign(loaded, loaded + nContigs);
        e->logMatch.resize(100); e->logMismatch.resize(100); makeQualityTables(e->logMatch.data(), e->logMismatch.data());
        std::memset(&e->R, 0, sizeof(e->R));
        e->R.bases = e->bases.data(); e->R.totalBases = e->bases.size(); e->R.contigOffset = e->offsets.data(); e->R.contigLoaded = e->loaded.data(); e->R.nContigs = nContigs;
        e->R.logMatch = e->logMatch.data(); e->R.logMismatch = e->logMismatch.data(); e->R.logStride = 1;
        std::memset(&e->cnt, 0, sizeof(e->cnt));
        return e;
    }
    catch (const std::exception &ex) { g_error = ex.what(); return 0; }
}
void emu_destroy(Emu *e) { delete e; }

// matches: grouped by cluster (ascending), any order inside a cluster, as isaac_gpu_find_matches delivers them; NoMatch records are dropped
int emu_set_matches(Emu *e, const isaac_match *m, u64 n, u32 nClusters)
{
    e->matches.clear(); e->matchOffsets.assign(size_t(nClusters) + 1, 0);
    u32 last = 0;
    for (u64 i = 0; i < n; ++i)
    {
        const u32 c = seedIdCluster(m[i].seed_id);
        if (c >= nClusters || c < last) { g_error = "matches must be grouped by ascending cluster"; return 1; }
        last = c;
        if (refposIsNoMatch(m[i].location)) continue;
        Match d; d.seedId = m[i].seed_id; d.location = m[i].location; e->matches.push_back(d);
        ++e->matchOffsets[c + 1];
    }
    for (u32 c = 0; c < nClusters; ++c) e->matchOffsets[c + 1] += e->matchOffsets[c];
    return 0;
}

int emu_build_fragments(Emu *e, const u8 *bcl, u32 nClusters, int withGaps, int trim,
                        isaac_candidate *out, u64 capacity, u64 *nOut, u32 *cigarOut, u64 cigarCapacity, u64 *nCigar)
{
    e->stores.resize(nClusters); e->frags.resize(nClusters);
    for (u32 c = 0; c < nClusters; ++c) e->frags[c] = e->stores[c].view();
    std::vector<FragmentWorkStore> workStore(1); std::vector<FragmentWork> work(1, workStore[0].bind());
    u64 generalKeysA[CAND_CAP], generalKeysB[CAND_CAP]; LeanKeyArea generalKeys; generalKeys.a = generalKeysA; generalKeys.b = generalKeysB; generalKeys.stride = 1;
    if (e->lean) { work[0].keys = &generalKeys; work[0].keyCap = CAND_CAP; }      // the general kernels sort on keys in LDS
    u64 n = 0, nc = 0;
    e->adapterRanges.assign(size_t(4) * nClusters + 4, 0);
    for (u32 c = 0; c < nClusters; ++c)
    {
        ClusterFragments &f = e->frags[c];
        // (the device addresses a cluster's ranges by its first candidate slot in the chunk's pool; here every cluster has a store of its own)
        if (e->P.adapters) { e->P.adapterCandBase = e->stores[c].cands; e->P.adapterRanges = &e->adapterRanges[size_t(4) * c]; }
        if (e->flatRescue && e->lean)
        {   // the kernels' sequence: lean forms for short lists (fragment_lean.h), the general ones for the rest
            u64 keysA[LEAN_LIST_MAX], keysB[LEAN_LIST_MAX];
            LeanKeyArea keys; keys.a = keysA; keys.b = keysB; keys.stride = 1;
            const u8 *clusterBcl = bcl + u64(c) * e->P.clusterLength;
            const u64 begin = e->matchOffsets[c], end = e->matchOffsets[c + 1];
            bool built;
            if (end - begin <= LEAN_LIST_MAX) built = leanBuildCandidates(e->P, clusterBcl, e->matches.data() + begin, u32(end - begin), trim != 0, f, keys);     // k_build_fragments
            else if (end - begin <= 160)
            {   // k_build_fragments_general
                u64 bigA[160], bigB[160]; u8 matchOrder[160], candOrder[160];
                LeanKeyArea big; big.a = bigA; big.b = bigB; big.stride = 1;
                built = keyedBuildCandidates(e->P, clusterBcl, e->matches.data() + begin, u32(end - begin), trim != 0, f, big, matchOrder, candOrder);
            }
            else built = buildCandidates(e->P, clusterBcl, e->matches.data() + begin, u32(end - begin), trim != 0, work[0], f);
            if (built)
            {
                if (e->P.adapters) for (u32 r = 0; r < e->P.nReads; ++r) for (u32 strand = 0; strand < 2; ++strand) clusterInitAdapterRanges(e->P, e->R, clusterBcl, f, r, strand);   // k_adapter_ranges
                for (u32 r = 0; r < e->P.nReads; ++r) for (u32 i = 0; i < f.nCands[r]; ++i) alignCandidate(e->P, e->R, clusterBcl, f, r, i, e->cnt);             // k_align_candidates
                if (f.nCands[0] <= LEAN_LIST_MAX && f.nCands[1] <= LEAN_LIST_MAX) leanFinishCandidates(e->P, f, keys);                                            // k_finish_candidates
                else finishCandidates(e->P, e->R, clusterBcl, work[0], f, e->cnt, true);
            }
            if (clusterSimpleIndelsPending(f)) clusterFinishSimpleIndels(e->P, e->R, bcl, c, work[0], f, e->cnt);   // k_indel_fragments
            const u32 nj = countGappedJobs(f, withGaps != 0);
            std::vector<GappedJob> jobs(nj + 1); std::vector<GappedResult> results(nj + 1);
            if (nj) writeGappedJobs(e->P, f, c, jobs.data());
            for (u32 j = 0; j < nj; ++j) runGappedJobSerial(e->P, e->R, bcl + u64(jobs[j].cluster) * e->P.clusterLength, jobs[j], work[0].tflags, results[j]);
            if (f.nCands[0] <= LEAN_LIST_MAX && f.nCands[1] <= LEAN_LIST_MAX)
            {   // k_finish_fragments
                u32 bswJobs = 0, bswAccepted = 0, candidates = 0;
                leanFinishFragments(e->P, f, nj ? results.data() : nullptr, keys, bswJobs, bswAccepted, candidates);
                e->cnt.bswJobs += bswJobs; e->cnt.bswAccepted += bswAccepted; e->cnt.candidates += candidates;
                if (f.flags & CLUSTER_OVERFLOW) ++e->cnt.overflowClusters;
            }
            else clusterFinishFragments(e->P, e->R, bcl, c, withGaps != 0, withGaps ? results.data() : nullptr, work[0], f, e->cnt);
        }
        else
        {
        clusterBuildFragments(e->P, e->R, bcl, c, e->matches.data(), e->matchOffsets.data(), withGaps != 0, trim != 0, work[0], f, e->cnt, e->flatRescue);
        if (clusterSimpleIndelsPending(f)) clusterFinishSimpleIndels(e->P, e->R, bcl, c, work[0], f, e->cnt);   // k_indel_fragments
        if (e->flatRescue && withGaps)
        {   // k_build_fragments -> k_gapped_jobs -> k_finish_fragments
            const u32 nj = countGappedJobs(f, true);
            std::vector<GappedJob> jobs(nj + 1); std::vector<GappedResult> results(nj + 1);
            if (nj) writeGappedJobs(e->P, f, c, jobs.data());
            for (u32 j = 0; j < nj; ++j) runGappedJobSerial(e->P, e->R, bcl + u64(jobs[j].cluster) * e->P.clusterLength, jobs[j], work[0].tflags, results[j]);
            clusterFinishFragments(e->P, e->R, bcl, c, true, results.data(), work[0], f, e->cnt);
        }
        else clusterFinishFragments(e->P, e->R, bcl, c, withGaps != 0, 0, work[0], f, e->cnt);
        }
        if (!out) continue;
        for (u32 r = 0; r < 2; ++r) for (u32 i = 0; i < f.nCands[r]; ++i)
        {
            const Cand &k = f.cands[r][i];
            if (n >= capacity || nc + k.cigarLength > cigarCapacity) { g_error = "capacity"; return 1; }
            isaac_candidate &o = out[n++]; std::memset(&o, 0, sizeof(o));
            o.position = k.position; o.log_probability = k.logProbability; o.cluster = c; o.read_index = k.readIndex; o.contig_id = k.contigId;
            o.observed_length = k.observedLength; o.reverse = k.reverse; o.mismatch_count = k.mismatchCount; o.matches_in_a_row = k.matchesInARow; o.gap_count = k.gapCount;
            o.edit_distance = k.editDistance; o.smith_waterman_score = k.smithWatermanScore; o.unique_seed_count = k.uniqueSeedCount;
            o.non_unique_first = k.nonUniqueFirst == NON_UNIQUE_NONE ? 0xffffffffu : k.nonUniqueFirst; o.non_unique_second = k.nonUniqueSecond;
            o.repeat_seeds_count = k.repeatSeedsCount; o.cigar_offset = u32(nc); o.cigar_length = k.cigarLength; o.low_clipped = k.lowClipped; o.high_clipped = k.highClipped;
            o.first_seed_index = k.firstSeedIndex;
            std::memcpy(cigarOut + nc, f.cigarPool + k.cigarOffset, k.cigarLength * 4); nc += k.cigarLength;
        }
    }
    if (nOut) *nOut = n; if (nCigar) *nCigar = nc;
    return 0;
}

int emu_determine_tls(Emu *e, const u8 *bcl, u32 nClusters, isaac_tls *out)
{
    int rc = emu_build_fragments(e, bcl, nClusters, 0, 0, 0, 0, 0, 0, 0, 0);
    if (rc) return rc;
    TlsLearner learner(e->P.mateDriftRange);
    if (2 == e->P.nReads)
    {
        for (u32 c = 0; c < nClusters && !learner.stats.stable; ++c)
        {
            TlsSample s; clusterTlsSample(e->frags[c], u32(e->matchOffsets[c + 1] - e->matchOffsets[c]), s);
            learner.add(s);
        }
        if (!learner.stats.stable) learner.finalize();
    }
    std::memcpy(out, &learner.stats, sizeof(*out));
    return 0;
}

int emu_select(Emu *e, const u8 *bcl, u32 nClusters, u32 tile, const isaac_tls *tls, isaac_fragment *records, u32 *cigars)
{
    int rc = emu_build_fragments(e, bcl, nClusters, 1, 1, 0, 0, 0, 0, 0, 0);
    if (rc) return rc;
    DevTls t; std::memcpy(&t, tls, sizeof(t));
    const RogCorrection rog = makeRogCorrection(e->P, e->offsets.data(), e->loaded.data(), e->R.nContigs);
    FragmentRecord *recs = reinterpret_cast<FragmentRecord *>(records);
    const double lmq40 = logMismatchQ40();
    auto makeWork = [](const TemplateCaps &caps, std::vector<u8> &arena, TemplateWork &work)
    {
        arena.assign(templateWorkBytes(caps) + 16, 0);
        templateWorkBind(work, reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(arena.data()) + 15) & ~uintptr_t(15)), caps);
    };
    std::vector<u8> lightArena, heavyArena; TemplateWork light, heavy;
    makeWork(lightCaps(), lightArena, light); makeWork(heavyCaps(), heavyArena, heavy);
    if (!e->flatRescue)
    {   // everything in the cluster's own thread (the round-1 baseline design)
        for (int tier = 0; tier < 2; ++tier)
            for (u32 c = 0; c < nClusters; ++c)
            {
                if (tier && !(recs[u64(c) * e->P.nReads].reserved & RECORD_TEMPLATE_OVERFLOW)) continue;
                { CoopInputs coop; coop.lanes = 1; coop.lane = 0; coop.fastSort = e->fastSort && tier; coop.ldsSort = 0; coop.ldsSortCap = 0;
                  clusterSelect(e->P, e->R, t, rog, lmq40, bcl, c, tile, e->frags[c], tier ? heavy : light, recs, cigars, e->cnt, 0, &coop); }
                if (!tier) ++e->cnt.clusters;
            }
        return 0;
    }
    // flat rescue: the same sequence of kernels as isaac_gpu_select, as plain loops
    // k_plan_rescue
    std::vector<u32> jobBase(nClusters + 1, 0);
    std::vector<RescueJob> jobs;
    // the lean form (template_lean.h) is what k_plan_rescue runs; the general one is kept for comparison (emu_set_lean)
    auto leanCtx = [&](u32 c, ClusterMeta &meta)
    {
        clusterViewStore(e->frags[c], e->stores[c].cands, meta);
        LeanCtx x; x.P = &e->P; x.R = &e->R; x.tls = &t;
        x.l0 = e->frags[c].cands[0]; x.l1 = e->frags[c].cands[1]; x.n0 = meta.nCands[0]; x.n1 = meta.nCands[1]; x.pool = e->frags[c].cigarPool;
        x.rogRead0 = rog.read[0]; x.rogRead1 = rog.read[1]; x.rog = rog.pair; x.logMismatchQ40 = lmq40; x.clusterId = c; x.mapqNearInteger = 0;
        return x;
    };
    auto plan = [&](u32 c, RescueJob *out) -> u32
    {
        if (!e->lean) return clusterPlanRescue(e->P, e->R, t, rog, lmq40, bcl, c, c, e->frags[c], light, out);
        if (!e->frags[c].built) return 0;
        ClusterMeta meta; LeanCtx x = leanCtx(c, meta);
        return leanPlanCluster(x, c, out);
    };
    for (u32 c = 0; c < nClusters; ++c) jobBase[c + 1] = jobBase[c] + plan(c, 0);
    jobs.resize(jobBase[nClusters]);
    for (u32 c = 0; c < nClusters; ++c) plan(c, jobs.data() + jobBase[c]);
    // k_rescue_windows
    std::vector<i32> candPositions;
    {
        Counters scratch; std::memset(&scratch, 0, sizeof(scratch));
        for (size_t j = 0; j < jobs.size(); ++j)
        {
            RescueJob &job = jobs[j];
            if (!job.valid) continue;
            ++e->cnt.rescueCalls; e->cnt.rescueWindowBases += job.windowLen;
            TemplateCtx x; templateCtxInit(x, e->P, e->R, t, rog, bcl, job.cluster, e->frags[job.cluster], heavy, scratch);
            const u32 n = findShadowCandidatePositions(x, e->R.bases + e->R.contigOffset[job.contigId], job.windowBegin, job.windowBegin + job.windowLen,
                                                       x.reads[job.shadowReadIndex], job.shadowReverse != 0);
            job.pushes = heavy.lastPushes + heavy.lastTruncated;
            if (heavy.lastTruncated) { job.fallback = 1; continue; }
            job.candBase = u32(candPositions.size()); job.nCands = n;
            for (u32 i = 0; i < n; ++i) candPositions.push_back(i32(heavy.candidatePositions[heavy.sortIdx[i]]));
            e->cnt.rescueCandidates += n;
        }
    }
    // k_rescue_adapter_ranges
    if (e->P.adapters)
        for (RescueJob &job : jobs)
        {
            if (!job.valid || job.fallback || !job.nCands) continue;
            ReadView shadowRead; const u32 r = job.shadowReadIndex;
            shadowRead.bcl = bcl + u64(job.cluster) * e->P.clusterLength + e->P.readOffset[r]; shadowRead.length = e->P.readLength[r]; shadowRead.firstCycle = e->P.firstCycle[r]; shadowRead.endCyclesMasked = 0;
            job.adapterRange = adapterStrandRange(*e->P.adapters, e->R, shadowRead, 0 != job.shadowReverse, job.contigId, i64(candPositions[job.candBase]) + job.windowBegin);
        }
    // k_rescue_align
    std::vector<Cand> shadowCands(candPositions.size()); std::vector<u32> shadowCigars(candPositions.size() * 3 + 3);
    std::vector<CandSummary> candSummaries(candPositions.size() + 1);          // the 16 bytes per candidate the plan kernels walk (k_rescue_align writes them)
    for (size_t j = 0; j < jobs.size(); ++j)
        for (u32 i = 0; i < jobs[j].nCands; ++i)
        {
            const u32 slot = jobs[j].candBase + i;
            rescueAlignCandidate(e->P, e->R, bcl, jobs[j].cluster, e->frags[jobs[j].cluster].endCyclesMasked[jobs[j].shadowReadIndex], jobs[j], candPositions[slot], shadowCands[slot], &shadowCigars[size_t(slot) * 3], &candSummaries[slot]);
            candSummaries[slot].relativePosition = rescueSummaryPosition(candPositions[slot], candSummaries[slot].cigarLength, shadowCigars[size_t(slot) * 3]);
            ++e->cnt.ungappedScans;
        }
    // k_rescue_gapped_plan + k_gapped_jobs
    std::vector<GappedResult> gapped; std::vector<u32> candRank(shadowCands.size() + 1);
    std::vector<GappedJob> gj;
    {
        std::vector<u32> tflags(3 * 512);
        for (size_t j = 0; j < jobs.size(); ++j)
        {
            RescueJob &job = jobs[j];
            if (!job.valid || job.fallback) continue;
            const u32 ecm = e->frags[job.cluster].endCyclesMasked[job.shadowReadIndex];
            summarizeRescueJob(job, shadowCands.data(), candRank.data(), candSummaries.data());
            const u32 n = job.nGapped;
            if (n != planRescueGapped(job, shadowCands.data(), shadowCigars.data(), ecm, 0)) { g_error = "summarizeRescueJob and planRescueGapped disagree"; return 1; }
            job.gappedBase = u32(gj.size()); job.nGapped = n;
            gj.resize(gj.size() + n);
            if (n && n != writeRescueGapped(job, shadowCands.data(), shadowCigars.data(), ecm, gj.data() + job.gappedBase, candSummaries.data())) { g_error = "summarizeRescueJob and writeRescueGapped disagree"; return 1; }
        }
        gapped.resize(gj.size() + 1);
        for (size_t j = 0; j < gj.size(); ++j) runGappedJobSerial(e->P, e->R, bcl + u64(gj[j].cluster) * e->P.clusterLength, gj[j], tflags.data(), gapped[j]);
    }
    e->dbgJobBase = jobBase; e->dbgJobs = jobs;
    // k_cluster_sums, then k_select on the precomputed results (private-memory work area); what either of them cannot do goes to
    // the wave-per-cluster pass with the reference's own capacities
    std::vector<u8> keyBytes(sumKeysBytes(1024) + 16);
    SumKeys keys; sumKeysBind(keys, keyBytes.data(), e->sumsCap ? e->sumsCap : 1024);
    std::vector<u8> tinyArena(templateWorkBytes(tinyCaps()) + 16, 0);
    TemplateWork tiny; templateWorkBind(tiny, reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(tinyArena.data()) + 15) & ~uintptr_t(15)), tinyCaps());
    for (u32 c = 0; c < nClusters; ++c)
    {
        if (e->P.adapters) { e->P.adapterCandBase = e->stores[c].cands; e->P.adapterRanges = &e->adapterRanges[size_t(4) * c]; }
        RescueInputs in; in.jobs = jobs.data() + jobBase[c]; in.jobCount = jobBase[c + 1] - jobBase[c]; in.shadowCands = shadowCands.data(); in.shadowCigars = shadowCigars.data();
        in.gappedResults = gapped.data(); in.gappedJobs = gj.data(); in.candRank = candRank.data(); in.sums = 0;
        const auto t0 = std::chrono::steady_clock::now();
        SumInputs si; si.jobs = in.jobs; si.nJobs = in.jobCount; si.shadowCands = shadowCands.data(); si.candRank = candRank.data(); si.gappedResults = gapped.data(); si.gappedJobs = gj.data(); si.shadowCigars = shadowCigars.data();
        SumGroup g; g.lanes = 1; g.lane = 0; g.block = false; g.radix = SumRadix{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; g.radixMin = 0; g.sumTile = nullptr; g.sumTileCap = 0;
        double sumTile[7]; if (e->sumsRadixMin >= 0) { g.sumTile = sumTile; g.sumTileCap = 7; }
        u16 radixCounts[16]; u32 radixTotals[1]; u64 radixVary[2]; std::vector<u16> radixAlt(1024); std::vector<u8> radixDigits(1024);
        if (e->sumsRadixMin >= 0) { g.radix.digits = (c & 1) ? radixDigits.data() : nullptr; g.radix.digitsCap = (c & 2) ? 1024 : 24; g.radix.counts = radixCounts; g.radix.totals = radixTotals; g.radix.vary = radixVary; g.radix.alt = radixAlt.data(); g.radixMin = u32(e->sumsRadixMin); }
        ClusterSums sums; u32 scratch = 0;
        bool residual = SUMS_DONE != clusterSums(e->P, e->frags[c], si, keys, g, &scratch, true, sums, e->cnt);
        CoopInputs coop; coop.lanes = 1; coop.lane = 0; coop.fastSort = false; coop.ldsSort = 0; coop.ldsSortCap = 0;
        if (!residual && e->lean)
        {   // k_select
            ClusterMeta meta; clusterViewStore(e->frags[c], e->stores[c].cands, meta);
            LeanRescue rs; rs.jobs = in.jobs; rs.jobCount = in.jobCount; rs.shadowCands = shadowCands.data(); rs.shadowCigars = shadowCigars.data(); rs.gappedResults = gapped.data(); rs.sums = &sums; leanOutcomesInPlace(rs);
            u32 near = 0;
            residual = !leanSelectCluster(e->P, e->R, t, rog, lmq40, bcl, c, tile, meta, e->stores[c].cands, e->frags[c].cigarPool, rs, recs, cigars, near);
            if (!residual) e->cnt.mapqNearInteger += near;
        }
        else if (!residual)
        {
            in.sums = &sums; in.serialFallbackAllowed = false;
            clusterSelect(e->P, e->R, t, rog, lmq40, bcl, c, tile, e->frags[c], tiny, recs, cigars, e->cnt, &in, &coop);
            residual = 0 != (recs[u64(c) * e->P.nReads].reserved & RECORD_TEMPLATE_OVERFLOW);
        }
        if (residual)
        {
            in.sums = 0; in.serialFallbackAllowed = true; coop.fastSort = e->fastSort;
            clusterSelect(e->P, e->R, t, rog, lmq40, bcl, c, tile, e->frags[c], heavy, recs, cigars, e->cnt, &in, &coop);
            ++e->cnt.heavyClusters;
        }
        if (e->clusterTimes) e->clusterTimes[c] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++e->cnt.clusters;
    }
    return 0;
}

// The candidate lists of lib/alignment/cppunit/testTemplateBuilder.cpp pushed through the product's template code (thread-serial
// form): ClusterFragments filled from literals, then clusterSelect.  Same record layout as tests/oracle capi's oracle_literal_fragment.
struct LiteralFragment
{
    u32 contigId; i64 position; u32 observedLength, readIndex, reverse, cigarOffset, cigarLength, mismatchCount;
    double logProbability; u32 uniqueSeedCount, alignmentScore, noMatch;
};
int emu_select_literal(Emu *e, const u8 *bcl, const LiteralFragment *f0, u32 n0, const LiteralFragment *f1, u32 n1, const isaac_tls *tls, isaac_fragment *records, u32 *cigars)
{
    if (n0 > CAND_CAP || n1 > CAND_CAP) { g_error = "too many candidates"; return 1; }
    std::vector<ClusterStore> stores(1);
    ClusterFragments f = stores[0].view();
    f.cands[1] = f.cands[0] + n0; f.candCap[1] = f.candCap[0] - n0;
    for (u32 k = 0; k < 16; ++k) f.cigarPool[k] = cigarOp(100, OP_ALIGN);     // cigarBuffer(1000, 1600) of the fixture
    f.cigarUsed = 16; f.built = (n0 || n1) ? 1 : 0;
    const LiteralFragment *in[2] = { f0, f1 }; const u32 n[2] = { n0, n1 };
    for (u32 r = 0; r < 2; ++r)
    {
        f.nCands[r] = n[r];
        for (u32 i = 0; i < n[r]; ++i)
        {
            const LiteralFragment &l = in[r][i];
            Cand &c = f.cands[r][i];
            candInit(c, l.readIndex);
            c.contigId = l.contigId; c.position = l.position; c.observedLength = l.observedLength; c.reverse = l.reverse != 0;
            c.cigarOffset = l.cigarOffset; c.cigarLength = u16(l.cigarLength); c.mismatchCount = u16(l.mismatchCount);
            c.logProbability = l.logProbability; c.uniqueSeedCount = u16(l.uniqueSeedCount); c.alignmentScore = l.alignmentScore;
        }
    }
    DevTls t; std::memcpy(&t, tls, sizeof(t));
    const RogCorrection rog = makeRogCorrection(e->P, e->offsets.data(), e->loaded.data(), e->R.nContigs);
    std::vector<u8> arena(templateWorkBytes(heavyCaps()) + 16, 0);
    TemplateWork work;
    templateWorkBind(work, reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(arena.data()) + 15) & ~uintptr_t(15)), heavyCaps());
    clusterSelect(e->P, e->R, t, rog, logMismatchQ40(), bcl, 0, 32, f, work, reinterpret_cast<FragmentRecord *>(records), cigars, e->cnt);
    return 0;
}

// realign.h (the BAM stage's gap realigner, thread-serial) on one fragment: the same inputs and outputs as the oracle's oracle_realign_case
int emu_realign_case(const char *contig, uint64_t contigLength, const uint8_t *readBcl, uint32_t readLength, uint64_t fStrandPosition, const uint32_t *cigar, uint32_t cigarLength,
                     uint32_t observedLength, uint32_t editDistance, uint32_t lowClipped, uint32_t highClipped, const int64_t *gapPositions, const int32_t *gapLengths, uint32_t nGaps,
                     uint32_t mismatchCost, uint32_t gapOpenCost, uint32_t gapExtendCost, int dodgy, int clipSemialigned, int vigorous, uint64_t binStart, int64_t binEnd,
                     uint64_t *realignedPosition, uint32_t *realignedCigar, uint32_t *realignedCigarLength, uint32_t *realignedEditDistance, uint32_t *realignedObservedLength)
{
    DevReference R; std::memset(&R, 0, sizeof(R));
    const u64 offsets[2] = { 0, contigLength };
    R.bases = contig; R.totalBases = contigLength; R.contigOffset = offsets; R.nContigs = 1;
    // RealignerGaps::finalizeGaps
    std::vector<RealignGap> gaps(nGaps);
    for (u32 i = 0; i < nGaps; ++i) { gaps[i].pos = refpos(0, u64(gapPositions[i])); gaps[i].length = gapLengths[i]; gaps[i].pad = 0; }
    std::sort(gaps.begin(), gaps.end(), [](const RealignGap &l, const RealignGap &r) { return rgLess(l, r); });
    gaps.erase(std::unique(gaps.begin(), gaps.end(), [](const RealignGap &l, const RealignGap &r) { return l.pos == r.pos && l.length == r.length; }), gaps.end());
    std::vector<RealignGap> ends;
    for (const RealignGap &g : gaps) if (rgIsDeletion(g)) ends.push_back(g);
    std::stable_sort(ends.begin(), ends.end(), [](const RealignGap &l, const RealignGap &r) { return rgEndPos(l, false) < rgEndPos(r, false); });
    RealignerGapsView view = { gaps.data(), u32(gaps.size()), ends.data(), u32(ends.size()) };
    RealignCtx x; x.R = &R; x.P.mismatchCost = mismatchCost; x.P.gapOpenCost = gapOpenCost; x.P.gapExtendCost = gapExtendCost; x.P.realignDodgyFragments = dodgy != 0; x.P.clipSemialigned = clipSemialigned != 0; x.P.realignGapsVigorously = vigorous != 0;
    RealignFragment f; std::memset(&f, 0, sizeof(f));
    f.fStrandPosition = refpos(0, fStrandPosition); f.mateFStrandPosition = f.fStrandPosition; f.observedLength = observedLength; f.flags = 0;
    f.lowClipped = u16(lowClipped); f.highClipped = u16(highClipped); f.alignmentScore = 1; f.templateAlignmentScore = 0; f.readLength = u16(readLength); f.editDistance = u16(editDistance); f.bcl = readBcl;
    RealignIndex index = { f.fStrandPosition, cigar, cigar + cigarLength };
    RealignCigar result; result.n = 0; result.overflow = false;
    const u64 binStartPos = refpos(0, binStart), binEndPos = binEnd < 0 ? refpos(0, binStart + contigLength) : refpos(0, u64(binEnd));
    realignFragment(x, view, binStartPos, binEndPos, index, f, result);
    *realignedPosition = refposPosition(index.pos);
    *realignedCigarLength = u32(index.cigarEnd - index.cigarBegin);
    for (const u32 *it = index.cigarBegin; it != index.cigarEnd; ++it) *realignedCigar++ = *it;
    *realignedEditDistance = f.editDistance; *realignedObservedLength = f.observedLength;
    return 0;
}

// A CPU model of the device deflate (deflate_kernels.h) for the table builder's sake: a serial greedy parse with the same kind of hash table,
// the very functions that make a token's bits (deflate_common.h) and the real makeDeflateTables.  One raw deflate block per call.
// counts (316 entries): in/out symbol statistics; tablesFromCounts: build the tables from `counts` as given, else from this input.
}
namespace isaac { bool makeDeflateTables(const u64 *litLenCounts, const u64 *distCounts, DeflateTables &t); }
extern "C" {
int emu_deflate(const u8 *data, u32 n, u64 *counts, int tablesFromCounts, u8 *out, u32 capacity, u32 *nOut)
{
    struct Token { u32 length, distance, byte; };
    std::vector<Token> tokens;
    std::vector<u32> table(1u << 13, 0xffffffffu);
    for (u32 p = 0; p < n;)
    {
        u32 length = 0, distance = 0;
        if (p + DEFLATE_MIN_MATCH <= n)
        {
            u32 w; std::memcpy(&w, data + p, 4);
            const u32 h = (w * 2654435761u) >> (32 - 13);
            const u32 candidate = table[h]; table[h] = p;
            if (candidate != 0xffffffffu && p - candidate <= DEFLATE_WINDOW && 0 == std::memcmp(data + candidate, data + p, 4))
            {
                const u32 limit = std::min(n - p, DEFLATE_MAX_MATCH);
                u32 l = 4; while (l < limit && data[candidate + l] == data[p + l]) ++l;
                length = l; distance = p - candidate;
            }
        }
        if (length) { tokens.push_back({ length, distance, 0 }); p += length; } else { tokens.push_back({ 0, 0, data[p] }); ++p; }
    }
    u64 own[DEFLATE_LITLEN_SYMBOLS + DEFLATE_DIST_SYMBOLS] = { 0 };
    for (const Token &t : tokens)
    {
        u32 xb, xv;
        if (t.length) { ++own[257 + deflateLengthCode(t.length, xb, xv)]; ++own[DEFLATE_LITLEN_SYMBOLS + deflateDistanceCode(t.distance, xb, xv)]; } else ++own[t.byte];
    }
    ++own[DEFLATE_END_OF_BLOCK];
    if (!tablesFromCounts) std::memcpy(counts, own, sizeof(own));
    DeflateTables tables;
    if (!makeDeflateTables(counts, counts + DEFLATE_LITLEN_SYMBOLS, tables)) { g_error = "header overflow"; return 1; }
    std::vector<u8> bytes; u64 acc = 0; u32 accBits = 0;
    auto put = [&](u64 bits, u32 nBits) { for (u32 b = 0; b < nBits; ++b) { acc |= ((bits >> b) & 1) << accBits; if (8 == ++accBits) { bytes.push_back(u8(acc)); acc = 0; accBits = 0; } } };
    for (u32 b = 0; b < tables.headerBits; ++b) put((tables.header[b >> 5] >> (b & 31)) & 1, 1);
    for (const Token &t : tokens)
    {
        u32 nBits; const u64 bits = t.length ? deflateMatchBits(tables.litLen, tables.dist, t.length, t.distance, nBits) : deflateLiteralBits(tables.litLen, t.byte, nBits);
        put(bits, nBits);
    }
    put(tables.litLen[DEFLATE_END_OF_BLOCK] & 0xffffu, tables.litLen[DEFLATE_END_OF_BLOCK] >> 16);
    if (accBits) bytes.push_back(u8(acc));
    *nOut = u32(bytes.size());
    if (bytes.size() > capacity) { g_error = "capacity"; return 1; }
    std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

void emu_set_flat_rescue(Emu *e, int on) { e->flatRescue = on != 0; }
void emu_set_lean(Emu *e, int on) { e->lean = on != 0; }
void emu_set_fast_sort(Emu *e, int on) { e->fastSort = on != 0; }
void emu_set_sums_radix(Emu *e, int minEntries) { e->sumsRadixMin = minEntries; }   // lists of at least this many entries are ordered by radixOrder (sums.h); -1: never
void emu_set_sums_capacity(Emu *e, uint32_t cap) { e->sumsCap = cap > 1024 ? 1024 : cap; }   // entries of the probability-sum key arrays (k_cluster_sums tiers)
void emu_set_cluster_times(Emu *e, double *t) { e->clusterTimes = t; }
// debugging: out = { jobs, valid jobs, total candidates, max candidates, gapped retries, fallback jobs, total window bases }
void emu_cluster_job_stats(Emu *e, u32 c, u64 *out)
{
    for (int i = 0; i < 7; ++i) out[i] = 0;
    if (c + 1 >= e->dbgJobBase.size()) return;
    for (u32 j = e->dbgJobBase[c]; j < e->dbgJobBase[c + 1]; ++j)
    {
        const RescueJob &job = e->dbgJobs[j];
        ++out[0]; out[1] += job.valid; out[2] += job.nCands; out[3] = std::max<u64>(out[3], job.nCands); out[4] += job.nGapped; out[5] += job.fallback; out[6] += job.windowLen;
    }
}

void emu_get_counters(Emu *e, isaac_counters *out) { std::memcpy(out, &e->cnt, sizeof(*out)); }

// banded SW leaf through the serial device function
int emu_bsw(int match, int mismatch, int gapOpen, int gapExtend, const char *query, u32 queryLength, const char *database, u32 *cigarOut, u32 *nOps, u32 *offset)
{
    DevParams P; std::memset(&P, 0, sizeof(P));
    P.gapMatch = match; P.gapMismatch = mismatch; P.gapOpen = -gapOpen; P.gapExtend = -gapExtend;
    std::vector<u32> tflags(3 * queryLength + 3);
    u32 words[256]; CigarPool pool; pool.words = words; pool.used = 0; pool.capacity = 256; pool.overflow = 0;
    struct Q { const char *q; char operator()(u32 i) const { return q[i]; } } q; q.q = query;
    *offset = bswAlignSerial(P, q, queryLength, database, tflags.data(), pool);
    *nOps = pool.used; std::memcpy(cigarOut, words, pool.used * 4);
    return 0;
}

// exactSort versus std::sort: sorts n records by key only and returns the permutation
void emu_exact_sort(const u32 *keys, u32 n, u16 *perm)
{
    for (u32 i = 0; i < n; ++i) perm[i] = u16(i);
    struct L { const u32 *k; bool operator()(u16 a, u16 b) const { return k[a] < k[b]; } } less; less.k = keys;
    exactSort(perm, int(n), less);
}
void emu_std_sort(const u32 *keys, u32 n, u16 *perm)
{
    for (u32 i = 0; i < n; ++i) perm[i] = u16(i);
    std::sort(perm, perm + n, [&](u16 a, u16 b) { return keys[a] < keys[b]; });
}

// the device logic of isaac_gpu_bam_records (bam_kernels.h: keys, record sizes, record bytes) with std::stable_sort in place of the radix passes
int emu_bam_records(const isaac_bam_tile *tiles, u32 nTiles, u32 nReads, const u32 *readLengths, u32 forcedDodgy, int pessimistic, const char *readGroup, const char *barcode,
                    u8 *out, u64 capacity, u64 *nBytes, u64 *nRecords, u64 *unalignedOffset)
{
    BamOptions o; std::memset(&o, 0, sizeof(o));
    o.nReads = nReads;
    for (u32 r = 0; r < nReads; ++r) { o.readLength[r] = readLengths[r]; o.readOffset[r] = o.clusterLength; o.clusterLength += readLengths[r]; }
    std::strcpy(o.readGroup, readGroup); o.readGroupLength = u32(std::strlen(readGroup)); std::strcpy(o.barcode, barcode); o.barcodeLength = u32(std::strlen(barcode));
    o.forcedDodgyAlignmentScore = forcedDodgy; o.pessimisticMapQ = pessimistic;
    std::vector<BamTile> t(nTiles);
    struct Key { u64 hi, lo; u32 tile; u64 index; };
    std::vector<Key> keys;
    for (u32 i = 0; i < nTiles; ++i)
    {
        std::memset(&t[i], 0, sizeof(BamTile));
        t[i].bcl = tiles[i].bcl_dev; t[i].records = reinterpret_cast<const FragmentRecord *>(tiles[i].fragments_dev); t[i].cigars = tiles[i].cigar_dev;
        t[i].nRecords = u32(tiles[i].n_records); t[i].nameLength = u32(std::strlen(tiles[i].read_name_prefix)); std::memcpy(t[i].name, tiles[i].read_name_prefix, t[i].nameLength);
        { const char *rg = tiles[i].read_group ? tiles[i].read_group : readGroup; t[i].readGroupLength = u32(std::strlen(rg)); std::memcpy(t[i].readGroup, rg, t[i].readGroupLength); }
        for (u64 k = 0; k < tiles[i].n_records; ++k)
        {
            const FragmentRecord &r = t[i].records[k];
            Key key; key.tile = i; key.index = k;
            key.hi = !bamStored(r) ? ~u64(0) : bamUnalignedBin(r) ? ~u64(0) - 1 : r.fStrandPosition;
            key.lo = ((u64(r.tile) * INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE + r.clusterId) << 2) | ((r.flags & 2) ? 2u : 0u) | ((r.flags & 64) ? 1u : 0u);
            keys.push_back(key);
        }
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); });
    u64 at = 0; *nRecords = 0; *unalignedOffset = ~u64(0);
    for (const Key &k : keys)
    {
        if (k.hi == ~u64(0)) break;
        if (k.hi == ~u64(0) - 1 && *unalignedOffset == ~u64(0)) *unalignedOffset = at;
        const FragmentRecord &r = t[k.tile].records[k.index];
        const u32 n = bamRecordBytes(t[k.tile], r, o);
        if (at + n <= capacity) { BamLayout l; bamLayout(t[k.tile], r, u64(&r - t[k.tile].records), o, l); if (l.total != n) return 9; BamStrings text = { t[k.tile].name, o.barcode, o.barcodeLength }; std::vector<u8> stored(l.readLength); for (u32 b = 0; b < l.readLength; ++b) stored[b] = bamStoredBcl(l, b); for (u32 j = 0; j < n; ++j) out[at + j] = bamRecordByte(text, l, j, (j & 1) ? stored.data() : nullptr, l.cigar); }
        at += n; ++*nRecords;
    }
    if (*unalignedOffset == ~u64(0)) *unalignedOffset = at;
    *nBytes = at;
    return at <= capacity ? 0 : 4;
}

// the CRC-32 arithmetic of k_bgzf_store (bgzf_kernels.h) in the kernel's own order: remainders of 260-byte pieces from a zero register, folded
// pairwise, the initial register value carried over the whole length at the end
uint32_t emu_crc32_folded(const u8 *data, u32 n)
{
    CrcConstants c; makeCrcConstants(c);
    std::vector<u32> partial(256, 0), lengths(256, 0);
    for (u32 t = 0; t < 256; ++t)
    {
        const u32 begin = t * 260, end = std::min(begin + 260, n);
        u32 crc = 0;
        u32 i = begin;
        for (; i + 4 <= end; i += 4)
        {
            u32 w; std::memcpy(&w, data + i, 4); w ^= crc;
            crc = c.table[3][w & 0xff] ^ c.table[2][(w >> 8) & 0xff] ^ c.table[1][(w >> 16) & 0xff] ^ c.table[0][w >> 24];
        }
        for (; i < end; ++i) crc = c.table[0][(crc ^ data[i]) & 0xff] ^ (crc >> 8);
        partial[t] = crc; lengths[t] = begin < n ? end - begin : 0;
    }
    for (u32 step = 1; step < 256; step <<= 1)
        for (u32 t = 0; t < 256; t += 2 * step)
            if (lengths[t + step]) { partial[t] = crcMultiply(partial[t], crcShiftOperator(c.squares, lengths[t + step])) ^ partial[t + step]; lengths[t] += lengths[t + step]; }
    return ~(crcMultiply(0xffffffffu, crcShiftOperator(c.squares, n)) ^ partial[0]);
}

uint32_t emu_sizeof(int what)
{
    switch (what) { case 0: return sizeof(Cand); case 1: return sizeof(ClusterMeta); case 2: return sizeof(FragmentWork); case 3: return sizeof(TemplateWork);
                    case 4: return sizeof(FragmentRecord); case 5: return sizeof(DevParams); case 6: return sizeof(isaac_params); default: return 0; }
}

} // extern "C"
