// Host side of the device deflate (deflate_kernels.h): the two Huffman code tables of a call and the dynamic block header that announces
// them (RFC 1951 3.2.7), from the symbol counts of a sample of the call's blocks.  Plain C++, no GPU, no zlib.
#include "deflate_common.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace isaac
{
namespace
{

// code lengths of a Huffman code for `counts` (every symbol with a non-zero count gets a code), none longer than maxBits.  Lengths beyond
// the limit are dealt with the simple way: the counts are flattened (halved, never to zero) and the tree is made again.
void huffmanLengths(const std::vector<u64> &countsIn, u32 maxBits, std::vector<u8> &lengths)
{
    const size_t n = countsIn.size();
    lengths.assign(n, 0);
    std::vector<u64> counts(countsIn);
    std::vector<u32> used;
    for (size_t i = 0; i < n; ++i) if (counts[i]) used.push_back(u32(i));
    if (used.empty()) return;
    if (1 == used.size()) { lengths[used[0]] = 1; return; }
    while (true)
    {
        // two-queue construction over the leaves sorted by count (ties by symbol, so that the result is deterministic)
        std::vector<u32> leaves(used);
        std::sort(leaves.begin(), leaves.end(), [&](u32 a, u32 b) { return counts[a] != counts[b] ? counts[a] < counts[b] : a < b; });
        const size_t m = leaves.size();
        std::vector<u64> weight(2 * m - 1); std::vector<u32> parent(2 * m - 1, 0);
        for (size_t i = 0; i < m; ++i) weight[i] = counts[leaves[i]];
        size_t leaf = 0, node = m, made = m;
        auto take = [&]() -> size_t
        {
            if (leaf < m && (node >= made || weight[leaf] <= weight[node])) return leaf++;
            return node++;
        };
        while (made < 2 * m - 1)
        {
            const size_t a = take(), b = take();
            weight[made] = weight[a] + weight[b]; parent[a] = u32(made); parent[b] = u32(made);
            ++made;
        }
        u32 longest = 0;
        std::vector<u32> depth(2 * m - 1, 0);
        for (size_t i = 2 * m - 2; i-- > 0;) depth[i] = depth[parent[i]] + 1;
        for (size_t i = 0; i < m; ++i) { lengths[leaves[i]] = u8(depth[i]); longest = std::max(longest, depth[i]); }
        if (longest <= maxBits) return;
        for (u32 s : used) counts[s] = (counts[s] + 1) / 2;
    }
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream; entry = code | length << 16
void canonicalCodes(const std::vector<u8> &lengths, u32 *out)
{
    u32 blCount[17] = { 0 }, nextCode[17] = { 0 };
    for (u8 l : lengths) ++blCount[l];
    blCount[0] = 0;
    u32 code = 0;
    for (u32 bits = 1; bits <= 16; ++bits) { code = (code + blCount[bits - 1]) << 1; nextCode[bits] = code; }
    for (size_t i = 0; i < lengths.size(); ++i)
    {
        const u32 l = lengths[i];
        if (!l) { out[i] = 0; continue; }
        const u32 c = nextCode[l]++;
        u32 r = 0;
        for (u32 b = 0; b < l; ++b) if (c & (1u << b)) r |= 1u << (l - 1 - b);
        out[i] = r | (l << 16);
    }
}

struct BitWriter
{
    u32 *words; u32 capacityWords; u32 bits = 0; bool overflow = false;
    void put(u32 value, u32 n)
    {
        for (u32 b = 0; b < n; ++b, ++bits)
        {
            if ((bits >> 5) >= capacityWords) { overflow = true; return; }
            if ((value >> b) & 1) words[bits >> 5] |= 1u << (bits & 31);
        }
    }
};

} // namespace

// litLenCounts: 286 entries (literals, end of block, length codes), distCounts: 30.  Every symbol gets a code whether it was seen or not
// (the counts come from a sample: a block outside it may use any symbol), unseen ones the longest.  Returns false when the header does not
// fit its words (cannot happen: see DEFLATE_HEADER_WORDS).
bool makeDeflateTables(const u64 *litLenCounts, const u64 *distCounts, DeflateTables &t)
{
    std::memset(&t, 0, sizeof(t));
    std::vector<u64> lit(litLenCounts, litLenCounts + DEFLATE_LITLEN_SYMBOLS), dist(distCounts, distCounts + DEFLATE_DIST_SYMBOLS);
    // seen symbols keep their proportions, unseen ones count as one occurrence in sixteen times the sample
    for (u64 &c : lit) c = c * 16 + 1;
    for (u64 &c : dist) c = c * 16 + 1;
    std::vector<u8> litLengths, distLengths;
    huffmanLengths(lit, 15, litLengths);
    huffmanLengths(dist, 15, distLengths);
    canonicalCodes(litLengths, t.litLen);
    canonicalCodes(distLengths, t.dist);
    // the code lengths of both tables as one sequence, run-length coded with 16 (repeat the previous length 3-6 times), 17 and 18 (3-10 / 11-138 zeros)
    std::vector<u8> all(litLengths); all.insert(all.end(), distLengths.begin(), distLengths.end());
    struct Item { u8 symbol; u8 extra; };
    std::vector<Item> items;
    for (size_t i = 0; i < all.size();)
    {
        size_t run = 1;
        while (i + run < all.size() && all[i + run] == all[i]) ++run;
        const u8 v = all[i];
        size_t left = run;
        if (0 == v)
        {
            while (left >= 11) { const size_t k = std::min<size_t>(left, 138); items.push_back({ 18, u8(k - 11) }); left -= k; }
            if (left >= 3) { items.push_back({ 17, u8(left - 3) }); left = 0; }
            while (left--) items.push_back({ 0, 0 });
        }
        else
        {
            items.push_back({ v, 0 }); --left;
            while (left >= 3) { const size_t k = std::min<size_t>(left, 6); items.push_back({ 16, u8(k - 3) }); left -= k; }
            while (left--) items.push_back({ v, 0 });
        }
        i += run;
    }
    std::vector<u64> clCounts(19, 0);
    for (const Item &it : items) ++clCounts[it.symbol];
    std::vector<u8> clLengths; u32 clCodes[19];
    huffmanLengths(clCounts, 7, clLengths);
    canonicalCodes(clLengths, clCodes);
    static const u8 order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    u32 hclen = 19;
    while (hclen > 4 && 0 == clLengths[order[hclen - 1]]) --hclen;
    BitWriter w; w.words = t.header; w.capacityWords = DEFLATE_HEADER_WORDS;
    w.put(1, 1);                                    // BFINAL: a BGZF block is one deflate block
    w.put(2, 2);                                    // BTYPE 10: dynamic Huffman codes
    w.put(DEFLATE_LITLEN_SYMBOLS - 257, 5); w.put(DEFLATE_DIST_SYMBOLS - 1, 5); w.put(hclen - 4, 4);
    for (u32 i = 0; i < hclen; ++i) w.put(clLengths[order[i]], 3);
    for (const Item &it : items)
    {
        w.put(clCodes[it.symbol] & 0xffffu, clCodes[it.symbol] >> 16);
        if (16 == it.symbol) w.put(it.extra, 2); else if (17 == it.symbol) w.put(it.extra, 3); else if (18 == it.symbol) w.put(it.extra, 7);
    }
    t.headerBits = w.bits;
    return !w.overflow;
}

} // namespace isaac
