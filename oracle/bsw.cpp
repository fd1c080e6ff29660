// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.hpp).
// Scalar, lane-by-lane restatement of the SSE2 banded Smith-Waterman of
// lib/alignment/BandedSmithWaterman.cpp:84-462.  All score arithmetic is done
// in wrapping int16 exactly as _mm_add_epi16/_mm_sub_epi16 do.
#include "oracle.hpp"
#include <stdexcept>
#include <algorithm>
#include <cstdlib>
#include <cassert>

namespace oracle
{

static inline int16_t w16(int v) { return int16_t(uint16_t(v)); }

// BandedSmithWaterman.cpp:36-54
BandedSmithWaterman::BandedSmithWaterman(int match, int mismatch, int gapOpen, int gapExtend, int maxLen)
    : matchScore(match), mismatchScore(mismatch), gapOpenScore(gapOpen), gapExtendScore(gapExtend), maxReadLength(maxLen),
      initialValue(int16_t(int(std::numeric_limits<short>::min()) + gapOpen)), T(size_t(maxLen) * 3 * 16)
{
    const int maxScore = std::max(std::max(std::max(std::abs(match), std::abs(mismatch)), std::abs(gapOpen)), std::abs(gapExtend));
    if ((maxLen * maxScore) >= std::abs(int(initialValue)))
        throw std::invalid_argument("BandedSmithWaterman: unsupported read length for these scores");
}

// the rows of the band, lane by lane (the restatement the AVX2 form below is checked against: ORACLE_BSW_SCALAR=1 selects it)
void BandedSmithWaterman::rowsScalar(const char *queryBegin, size_t querySize, const char *dbBegin, int16_t *G, int16_t *E, int16_t *F) const
{
    const int16_t open = int16_t(gapOpenScore), ext = int16_t(gapExtendScore);
    (void)ext;
    for (unsigned k = 0; k < 16; ++k) { E[k] = initialValue; F[k] = 0; G[k] = initialValue; } // :109-114 (F = 0 quirk)
    G[0] = 0;                                                                                  // :115
    uint8_t *t = &T[0];
    for (unsigned i = 0; i < querySize; ++i)
    {
        int16_t newF[16], newG[16];
        uint8_t TF[16], TG[16], TE[16];
        // F: :130-173
        for (unsigned k = 1; k < 16; ++k)
        {
            const int16_t g = G[k - 1], e = E[k - 1];
            uint8_t tf = (g < e) ? 1 : 0;                       // :142-145
            int16_t v = w16(std::max(g, e) - open);             // :150-151 (wrapping sub)
            const int16_t fe = w16(F[k - 1] - ext);             // :154-158
            if (v < fe) tf = std::max<uint8_t>(2, tf);          // :162-166 (_mm_max_epu8)
            newF[k] = std::max(v, fe);                          // :171
            TF[k] = tf;
        }
        newF[0] = initialValue; TF[0] = 0;                      // :167,173
        // G: :174-197
        uint8_t fE[16], fF[16];
        for (unsigned k = 0; k < 16; ++k)
        {
            fE[k] = (G[k] < E[k]) ? 1 : 0;
            const int16_t m = std::max(G[k], E[k]);
            fF[k] = (m < F[k]) ? 2 : 0;
            newG[k] = std::max(m, F[k]);
        }
        // :197  _mm_max_epi16 over byte pairs (even lane = low byte, odd lane = high byte)
        for (unsigned m = 0; m < 8; ++m)
        {
            const int16_t a = int16_t(uint16_t(fF[2 * m]) | (uint16_t(fF[2 * m + 1]) << 8));
            const int16_t b = int16_t(uint16_t(fE[2 * m]) | (uint16_t(fE[2 * m + 1]) << 8));
            const uint16_t r = uint16_t(std::max(a, b));
            TG[2 * m] = uint8_t(r & 0xff); TG[2 * m + 1] = uint8_t(r >> 8);
        }
        // W: :200-244 -- byte compare of query base against database byte of the lane
        const char q = queryBegin[i];
        for (unsigned k = 0; k < 16; ++k)
        {
            const char d = dbBegin[i + 15 - k];
            const bool diff = (q != d);
            // unpack(W,B): low byte = score byte, high byte = 0xff on mismatch, 0 on match
            const uint8_t wb = diff ? uint8_t(mismatchScore) : uint8_t(matchScore);
            const uint16_t w = uint16_t(wb) | (diff ? 0xff00 : 0);
            newG[k] = w16(int(newG[k]) + int(int16_t(w)));
        }
        // E: :246-297, serial from lane 15 down to lane 0
        {
            int16_t g = initialValue, e = initialValue, f = initialValue;
            for (unsigned j = 0; j < 16; ++j)
            {
                const unsigned k = 15 - j;
                int16_t mx = g; uint8_t tMax = 0;
                if (e > g && e > f) { mx = e; tMax = 1; }
                else if (f > g) { mx = f; tMax = 2; }
                TE[k] = tMax;
                E[k] = mx;
                g = w16(newG[k] - open);
                e = w16(int(mx) - gapExtendScore);
                f = w16(newF[k] - open);
            }
        }
        for (unsigned k = 0; k < 16; ++k) { G[k] = newG[k]; F[k] = newF[k]; }
        for (unsigned k = 0; k < 16; ++k) { t[k] = TG[k]; t[16 + k] = TE[k]; t[32 + k] = TF[k]; } // :306-308
        t += 48;
    }
}

// The same rows sixteen lanes at a time (AVX2: the whole band of 16 x int16 in one register, where the reference's SSE2 code holds it in two): F, G, the
// 16-bit max over pairs of flag bytes and W as the reference's vector statements; the E chain, which the reference also walks lane by lane (:246-297),
// on the lanes stored to memory.  Checked against rowsScalar on the reference's 301 known-answer cases and on random ones (tests/test_oracle_golden.py).
static thread_local int g_forceScalarRows = 0;
void bswForceScalarRows(int on) { g_forceScalarRows = on; }
bool BandedSmithWaterman::useAvx2()
{
#if defined(__AVX2__)
    static const bool scalar = 0 != std::getenv("ORACLE_BSW_SCALAR");
    return !scalar && !g_forceScalarRows;
#else
    return false;
#endif
}
#if defined(__AVX2__)
} // namespace oracle
#include <immintrin.h>
namespace oracle
{
void BandedSmithWaterman::rowsAvx2(const char *queryBegin, size_t querySize, const char *dbBegin, int16_t *Gout, int16_t *Eout, int16_t *Fout) const
{
    const __m256i open = _mm256_set1_epi16(int16_t(gapOpenScore)), ext = _mm256_set1_epi16(int16_t(gapExtendScore));
    const __m256i init = _mm256_set1_epi16(initialValue), one = _mm256_set1_epi16(1), two = _mm256_set1_epi16(2);
    const __m256i lane0 = _mm256_setr_epi16(-1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    alignas(32) int16_t e0[16], g0[16];
    for (unsigned k = 0; k < 16; ++k) { e0[k] = initialValue; g0[k] = initialValue; }
    g0[0] = 0;
    __m256i E = _mm256_load_si256(reinterpret_cast<const __m256i *>(e0)), F = _mm256_setzero_si256(), G = _mm256_load_si256(reinterpret_cast<const __m256i *>(g0));
    const __m128i reverse = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i wMatch = _mm256_set1_epi16(int16_t(uint16_t(uint8_t(matchScore)))), wMismatch = _mm256_set1_epi16(int16_t(uint16_t(0xff00u | uint8_t(mismatchScore))));
    const int16_t openS = int16_t(gapOpenScore);
    uint8_t *t = &T[0];
    alignas(32) int16_t gmo[16], fmo[16], eNew[16];
    for (size_t i = 0; i < querySize; ++i)
    {
        // F (:130-173): computed where its inputs are, then moved up a lane; lane 0 gets the initial value and flag 0
        const __m256i m = _mm256_max_epi16(G, E);
        const __m256i gltE = _mm256_cmpgt_epi16(E, G);                                   // G < E
        const __m256i v = _mm256_sub_epi16(m, open), fe = _mm256_sub_epi16(F, ext);
        const __m256i vltFe = _mm256_cmpgt_epi16(fe, v);
        const __m256i tfHere = _mm256_blendv_epi8(_mm256_and_si256(gltE, one), two, vltFe);   // v < fe ? 2 : (G < E ? 1 : 0)
        const __m256i nfHere = _mm256_max_epi16(v, fe);
        const __m256i t8 = _mm256_permute2x128_si256(nfHere, nfHere, 0x08), t8f = _mm256_permute2x128_si256(tfHere, tfHere, 0x08);
        __m256i newF = _mm256_alignr_epi8(nfHere, t8, 14), TF = _mm256_alignr_epi8(tfHere, t8f, 14);
        newF = _mm256_blendv_epi8(newF, init, lane0); TF = _mm256_andnot_si256(lane0, TF);
        // G (:174-197) and its flags: fE = G < E, fF = max(G, E) < F
        const __m256i fE = _mm256_and_si256(gltE, one);
        const __m256i fF = _mm256_and_si256(_mm256_cmpgt_epi16(F, m), two);
        __m256i newG = _mm256_max_epi16(m, F);
        // :197 the 16-bit max over pairs of flag bytes: the flags as sixteen bytes, viewed as eight int16
        const __m256i packedF = _mm256_permute4x64_epi64(_mm256_packus_epi16(fF, fF), 0xd8), packedE = _mm256_permute4x64_epi64(_mm256_packus_epi16(fE, fE), 0xd8);
        const __m128i TG = _mm_max_epi16(_mm256_castsi256_si128(packedF), _mm256_castsi256_si128(packedE));
        // W (:200-244): byte compare of the row's query base against the database bytes of the lanes (lane k: database[i + 15 - k])
        const __m128i d = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(dbBegin + i)), reverse);
        const __m128i same = _mm_cmpeq_epi8(d, _mm_set1_epi8(queryBegin[i]));
        const __m256i same16 = _mm256_cvtepi8_epi16(same);                               // 0xffff where equal
        newG = _mm256_add_epi16(newG, _mm256_blendv_epi8(wMismatch, wMatch, same16));
        // E (:246-297): serial from lane 15 down to lane 0
        _mm256_store_si256(reinterpret_cast<__m256i *>(gmo), _mm256_sub_epi16(newG, open));
        _mm256_store_si256(reinterpret_cast<__m256i *>(fmo), _mm256_sub_epi16(newF, open));
        {
            int16_t g = initialValue, e = initialValue, f = initialValue;
            for (unsigned j = 0; j < 16; ++j)
            {
                const unsigned k = 15 - j;
                int16_t mx = g; uint8_t tMax = 0;
                if (e > g && e > f) { mx = e; tMax = 1; }
                else if (f > g) { mx = f; tMax = 2; }
                t[16 + k] = tMax;
                eNew[k] = mx;
                g = gmo[k]; e = w16(int(mx) - gapExtendScore); f = fmo[k];
            }
        }
        (void)openS;
        E = _mm256_load_si256(reinterpret_cast<const __m256i *>(eNew));
        G = newG; F = newF;
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), TG);
        const __m256i packedTF = _mm256_permute4x64_epi64(_mm256_packus_epi16(TF, TF), 0xd8);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t + 32), _mm256_castsi256_si128(packedTF));
        t += 48;
    }
    _mm256_store_si256(reinterpret_cast<__m256i *>(Gout), G); _mm256_store_si256(reinterpret_cast<__m256i *>(Eout), E); _mm256_store_si256(reinterpret_cast<__m256i *>(Fout), F);
}
#else
void BandedSmithWaterman::rowsAvx2(const char *queryBegin, size_t querySize, const char *dbBegin, int16_t *G, int16_t *E, int16_t *F) const { rowsScalar(queryBegin, querySize, dbBegin, G, E, F); }
#endif

// BandedSmithWaterman.cpp:84-462.  Lane k (0..15) of row i corresponds to database index i + 15 - k.
unsigned BandedSmithWaterman::align(const char *queryBegin, const char *queryEnd, const char *dbBegin, const char *dbEnd, Cigar &cigar) const
{
    const size_t querySize = queryEnd - queryBegin;
    assert(querySize + WIDEST_GAP_SIZE - 1 == size_t(dbEnd - dbBegin));
    assert(querySize <= size_t(maxReadLength));
    (void)dbEnd;
    const size_t originalCigarSize = cigar.size();
    alignas(32) int16_t E[16], F[16], G[16];
    if (useAvx2()) rowsAvx2(queryBegin, querySize, dbBegin, G, E, F);
    else rowsScalar(queryBegin, querySize, dbBegin, G, E, F);
    // :349-379 end-cell scan
    int16_t mx = w16(int(uint16_t(G[15])) - 1);
    int ii = int(querySize) - 1;
    int jj = ii;
    unsigned maxType = 0;
    const int16_t *TT[3] = { G, E, F };
    for (int k = 15; k >= 0; --k)
        for (unsigned type = 0; type < 3; ++type)
        {
            const int16_t value = TT[type][k];
            if (value > mx) { mx = value; jj = k; maxType = type; }
        }
    // :381-435 traceback
    static const int jjIncrement[3] = { 0, 1, -1 };
    static const int iiIncrement[3] = { -1, 0, -1 };
    static const CigarOp opCodes[3] = { ALIGN, DELETE, INSERT };
    unsigned opLength = 0;
    if (jj > 0) cigar.push_back(cigarEncode(jj, DELETE));
    while (ii >= 0 && jj >= 0 && jj <= 15)
    {
        ++opLength;
        const unsigned nextMaxType = T[(size_t(ii) * 3 + maxType) * 16 + jj];
        if (nextMaxType != maxType) { cigar.push_back(cigarEncode(opLength, opCodes[maxType])); opLength = 0; }
        ii += iiIncrement[maxType];
        jj += jjIncrement[maxType];
        maxType = nextMaxType;
    }
    assert(-1 == ii);
    if (1 != maxType && opLength) { cigar.push_back(cigarEncode(opLength, opCodes[maxType])); opLength = 0; }
    if (15 > jj) { cigar.push_back(cigarEncode(opLength + 15 - jj, DELETE)); opLength = 0; }
    assert(0 == opLength);
    // :437-453 strip leading/trailing deletions
    unsigned ret = 0;
    const std::pair<unsigned, CigarOp> first = cigarDecode(cigar.back());
    if (DELETE == first.second) { cigar.pop_back(); ret = first.first; }
    std::reverse(cigar.begin() + originalCigarSize, cigar.end());
    if (DELETE == cigarDecode(cigar.back()).second) cigar.pop_back();
    return ret;
}

} // namespace oracle
