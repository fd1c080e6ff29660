// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  CPU restatement of the BAM side of the path for
// --realign-gaps no | sample, --mark-duplicates 0 | 1, --keep-duplicates 0 | 1 and the default tag set (--bam-exclude-tags ZX,ZY):
//   gap realignment (--realign-gaps sample)    lib/build/BinSorter.cpp:387-417 around realign.cpp (GapRealigner), every contig one bin
//   duplicate marking                          lib/build/BinSorter.cpp:293-330, include/build/DuplicateFragmentIndexFiltering.hh:37-208,
//                                              include/build/DuplicatePairEndFilter.hh:45-107, include/io/Fragment.hh:66-71,490-506
//   what FragmentCollector keeps per read      lib/alignment/matchSelector/FragmentCollector.cpp:43-111
//   the order of a bin                         include/build/PackedFragmentBuffer.hh:149-176 (orderForBam), lib/build/BinSorter.cpp
//   the record adapter                         include/build/FragmentAccessorBamAdapter.hh:127-377
//   record and header serialisation            include/bam/Bam.hh:147-345, lib/bam/Bam.cpp:38-45
// Bins follow each other in position order and the unaligned bin is written last (--keep-unaligned back), so one sort over
// all aligned records and shadows followed by the unaligned templates in storage order is the record stream of the file.
#include "oracle.hpp"

#include <algorithm>
#include <map>
#include <cstring>
#include <string>

namespace oracle
{
namespace
{

const uint64_t CLUSTERS_PER_TILE_FACTOR = 1000000000UL;     // include/build/FragmentIndex.hh INSANELY_HIGH_NUMBER_OF_CLUSTERS_PER_TILE
const uint16_t DODGY = 0xffff;
const uint64_t NO_MATCH_VALUE = ReferencePosition(ReferencePosition::NoMatch).value;

// the stored form of one read: header fields, bases as stored (FragmentCollector::storeBclAndCigar), CIGAR
struct Stored
{
    FragmentRecord *header; std::vector<unsigned char> bases; const uint32_t *cigarBegin, *cigarEnd; const std::string *namePrefix; const std::string *readGroup = 0; const TemplateLengthStatistics *tls = 0;
    const uint32_t *originalCigarBegin = 0, *originalCigarEnd = 0;                     // FragmentAccessorBamAdapter::originalCigarBegin_: the fragment's own CIGAR
    const uint8_t *clusterBcl = 0; FragmentRecord *mate = 0; bool duplicate = false;
    bool paired() const { return header->flags & 1; }
    bool unmapped() const { return header->flags & 2; }
    bool mateUnmapped() const { return header->flags & 4; }
    bool reverse() const { return header->flags & 8; }
    bool secondRead() const { return header->flags & 64; }
    bool properPair() const { return header->flags & 256; }
};

inline bool isBclN(unsigned char b) { return !(b & 0xfc); }                                   // oligo/Nucleotides.hh:91-94
inline unsigned char reverseBcl(unsigned char b) { return !isBclN(b) ? (b & 0xfc) | (3 - (b & 3)) : 0; }   // :153-156

void put(std::vector<char> &os, const void *p, size_t n) { const char *c = static_cast<const char *>(p); os.insert(os.end(), c, c + n); }
void putInt(std::vector<char> &os, int v) { put(os, &v, 4); }
void putUnsigned(std::vector<char> &os, unsigned v) { put(os, &v, 4); }

int reg2bin(unsigned beg, unsigned end)                                                       // Bam.hh:237-246
{
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

struct Adapter                                                                                 // FragmentAccessorBamAdapter
{
    const Stored &s; uint64_t pos; bool withCigar; const BamOptions &o;
    bool noMatch() const { return pos == NO_MATCH_VALUE; }
    int refId() const { return noMatch() ? -1 : int(ReferencePosition::fromValue(pos).getContigId()); }
    int position() const { return noMatch() ? -1 : int(ReferencePosition::fromValue(pos).getPosition()); }
    std::string readName() const { return *s.namePrefix + std::to_string(s.header->clusterId) + ":0"; }
    unsigned char mapq() const
    {
        const FragmentRecord &h = *s.header;
        if (s.properPair())
        {
            if (DODGY == h.templateAlignmentScore) return o.forcedDodgyAlignmentScore;
            return std::min<unsigned>(60U, o.pessimisticMapQ ? std::min(h.alignmentScore, h.templateAlignmentScore) : std::max(h.alignmentScore, h.templateAlignmentScore));
        }
        return DODGY == h.alignmentScore ? o.forcedDodgyAlignmentScore : std::min<unsigned>(60U, h.alignmentScore);
    }
    unsigned flag() const
    {
        unsigned bs = 0;
        bs |= unsigned(s.paired()) << 0; bs |= unsigned(s.properPair()) << 1; bs |= unsigned(s.unmapped()) << 2; bs |= unsigned(s.paired() && s.mateUnmapped()) << 3;
        bs |= unsigned(s.reverse()) << 4; bs |= unsigned(bool(s.header->flags & 16)) << 5; bs |= unsigned(s.paired() && (s.header->flags & 32)) << 6;
        bs |= unsigned(s.paired() && s.secondRead()) << 7; bs |= unsigned(bool(s.header->flags & 128)) << 9;
        bs |= unsigned(s.duplicate) << 10;                                                       // FragmentAccessorBamAdapter.hh:357
        return bs;
    }
    int nextRefId() const { return s.paired() ? (s.unmapped() && s.mateUnmapped() ? -1 : int(ReferencePosition::fromValue(s.header->mateFStrandPosition).getContigId())) : -1; }
    int nextPos() const { return s.paired() ? (s.unmapped() && s.mateUnmapped() ? -1 : int(ReferencePosition::fromValue(s.header->mateFStrandPosition).getPosition())) : -1; }
};

unsigned char bamBase(unsigned char b) { return !isBclN(b) ? 1 << (b & 3) : 15; }              // bamBaseFromBclByte

// bam::serializeAlignment
void serializeAlignment(std::vector<char> &os, const Adapter &a)
{
    const FragmentRecord &h = *a.s.header;
    const int refID = a.refId(), pos = a.position();
    const std::string name = a.readName();
    const unsigned observedLength = h.observedLength;
    const unsigned bin_mq_nl = unsigned(reg2bin(pos, pos + (observedLength ? observedLength : 1))) << 16 | unsigned(a.mapq()) << 8 | unsigned(name.size() + 1);
    const size_t cigarLength = a.withCigar ? a.s.cigarEnd - a.s.cigarBegin : 0;
    const unsigned flag_nc = a.flag() << 16 | (unsigned(cigarLength) & 0xFFFF);
    const int l_seq = h.readLength;
    std::vector<unsigned char> seq((l_seq + 1) / 2, 15), qual;
    for (int i = 0; i + 1 < l_seq; i += 2) seq[i / 2] = bamBase(a.s.bases[i]) << 4 | bamBase(a.s.bases[i + 1]);
    if (l_seq % 2) seq[l_seq / 2] = bamBase(a.s.bases[l_seq - 1]) << 4;
    for (int i = 0; i < l_seq; ++i) qual.push_back(a.s.bases[i] >> 2);
    const bool sm = DODGY != h.alignmentScore, as = a.s.properPair() && DODGY != h.templateAlignmentScore;
    // getFragmentOC (FragmentAccessorBamAdapter.hh:182-198): the CIGAR before realignment, for fragments whose CIGAR now lies in the realigner's buffer
    std::string oc;
    const bool realigned = a.s.cigarBegin != a.s.originalCigarBegin;
    if (realigned) for (const uint32_t *c = a.s.originalCigarBegin; c != a.s.originalCigarEnd; ++c) { oc += std::to_string(*c >> 4); oc += "MIDNSHP=X?"[std::min<uint32_t>(*c & 15, 9)]; }
    const int block_size = 32 + int(name.size()) + 1 + int(cigarLength) * 4 + int(seq.size()) + int(qual.size()) + (sm ? 7 : 0) + (as ? 7 : 0) + 7 +
                           (3 + int(a.o.barcode.size()) + 1) + (3 + int(a.s.readGroup->size()) + 1) + (realigned ? 3 + int(oc.size()) + 1 : 0);
    putInt(os, block_size); putInt(os, refID); putInt(os, pos); putUnsigned(os, bin_mq_nl); putUnsigned(os, flag_nc);
    putInt(os, l_seq); putInt(os, a.nextRefId()); putInt(os, a.nextPos()); putInt(os, h.bamTlen);
    put(os, name.c_str(), name.size() + 1);
    put(os, a.s.cigarBegin, cigarLength * 4);
    put(os, seq.data(), seq.size()); put(os, qual.data(), qual.size());
    if (sm) { put(os, "SMi", 3); putInt(os, h.alignmentScore); }
    if (as) { put(os, "ASi", 3); putInt(os, h.templateAlignmentScore); }
    put(os, "RGZ", 3); put(os, a.s.readGroup->c_str(), a.s.readGroup->size() + 1);
    put(os, "NMi", 3); putInt(os, h.editDistance);
    put(os, "BCZ", 3); put(os, a.o.barcode.c_str(), a.o.barcode.size() + 1);
    if (realigned) { put(os, "OCZ", 3); put(os, oc.c_str(), oc.size() + 1); }
}

// oligo::pack32BclBases (include/oligo/Nucleotides.hh:241-280)
uint64_t pack32BclBases(const uint8_t *bcl, unsigned available)
{
    uint64_t ret = 0;
    for (unsigned i = 0; i < 32 && i < available; ++i) ret |= uint64_t(bcl[i] & 3) << (2 * i);
    return ret;
}
// ReferencePosition value of the position `offset` bases further on the same contig
uint64_t advance(uint64_t positionValue, uint64_t offset) { return positionValue + (offset << 1); }
// io::FragmentIndexAnchor (Fragment.hh:490-506): aligned reads are anchored at their lowest cycle (f-strand position of forward reads, r-strand
// position of reverse ones), shadows at their first 32 bases
uint64_t anchorOf(const FragmentRecord &h, const uint8_t *readBcl)
{
    if (!(h.flags & 2)) return (h.flags & 8) ? advance(h.fStrandPosition, std::max(h.observedLength, 1U) - 1) : h.fStrandPosition;
    return pack32BclBases(readBcl, h.readLength);
}

} // namespace

// FDuplicateFilter::less / RSDuplicateFilter::less (DuplicateFragmentIndexFiltering.hh:42-88,126-175)
static bool duplicateLess(const PairEndIndex &left, const PairEndIndex &right)
{
    if (left.primary < right.primary) return true;
    if (left.primary == right.primary)
    {
        if (left.mateAnchor < right.mateAnchor) return true;
        if (left.mateAnchor == right.mateAnchor)
        {
            if (left.mateInfo < right.mateInfo) return true;
            if (left.mateInfo == right.mateInfo)
            {
                if (left.library < right.library) return true;
                if (left.library == right.library)
                {
                    if (left.duplicateClusterRank > right.duplicateClusterRank) return true;      // higher alignment score on top
                    if (left.duplicateClusterRank == right.duplicateClusterRank && left.globalClusterId < right.globalClusterId) return true;
                }
            }
        }
    }
    return false;
}
// ::equal_to (:89-115,176-206): both ends of one cluster are never duplicates of each other
static bool duplicateEqual(const PairEndIndex &left, const PairEndIndex &right)
{
    if (left.primary == right.primary && left.mateAnchor == right.mateAnchor && left.mateInfo == right.mateInfo)
        if (left.globalClusterId != right.globalClusterId) return left.library == right.library;
    return false;
}
void filterDuplicates(std::vector<PairEndIndex> &ends, std::vector<char> &isDuplicate)
{
    isDuplicate.assign(ends.size(), 0);
    if (ends.empty()) return;
    std::sort(ends.begin(), ends.end(), duplicateLess);
    size_t last = 0;
    for (size_t it = 1; it < ends.size(); ++it)
    {
        if (!duplicateEqual(ends[last], ends[it])) last = it;
        else isDuplicate[it] = 1;
    }
}

void bamRecords(const std::vector<BamTileInput> &tiles, const BamOptions &o, std::vector<char> &os, uint64_t &nRecords, uint64_t &unalignedOffset)
{
    // the realigner changes records: everything below works on copies
    std::vector<std::vector<FragmentRecord> > copies;
    for (const BamTileInput &t : tiles) copies.push_back(std::vector<FragmentRecord>(t.records, t.records + t.nRecords));
    std::vector<uint32_t> realignedCigars;
    std::vector<Stored> stored;
    for (size_t tileIndex = 0; tileIndex < tiles.size(); ++tileIndex)
    {
        const BamTileInput &t = tiles[tileIndex];
        for (uint64_t i = 0; i < t.nRecords; ++i)
        {
            FragmentRecord &h = copies[tileIndex][i];
            if (h.reserved & 2) continue;                                     // MatchSelector.cpp:345-357: the template was not stored
            Stored s; s.header = &h; s.namePrefix = &t.namePrefix; s.readGroup = t.readGroup.empty() ? &o.readGroup : &t.readGroup; s.tls = t.tls ? t.tls : o.tls;
            const unsigned readIndex = (h.flags & 1) && (h.flags & 64) ? 1 : 0;
            const uint8_t *bcl = t.bcl + uint64_t(h.clusterId) * o.clusterLength + o.readOffset[readIndex];
            s.bases.assign(bcl, bcl + h.readLength);
            if (s.reverse()) { std::reverse(s.bases.begin(), s.bases.end()); for (unsigned char &b : s.bases) b = reverseBcl(b); }
            s.cigarBegin = t.cigars + h.cigarOffset; s.cigarEnd = s.cigarBegin + ((h.flags & 2) ? 0 : h.cigarLength);
            s.originalCigarBegin = s.cigarBegin; s.originalCigarEnd = s.cigarEnd;
            s.clusterBcl = t.bcl + uint64_t(h.clusterId) * o.clusterLength;
            if (h.flags & 1) s.mate = &copies[tileIndex][i ^ 1];              // records come in cluster order, read 0 before read 1
            stored.push_back(s);
        }
    }
    // the ends of pairs in the order of the bin's index after BinSorter::resolveDuplicates: reverse-strand ends and shadows, then forward-strand
    // ends, each list as the duplicate filter's sort leaves it (without filtering the reference keeps the order of its bin files, which is the
    // order its threads happened to store fragments in; the sorted order stands in for it)
    // the bin of a position: its contig and the number of cuts at or before it (cuts of earlier contigs count for all positions of this one alike)
    const auto cutsUpTo = [&o](uint64_t value) { return uint64_t(std::upper_bound(o.binCuts.begin(), o.binCuts.end(), value & ~uint64_t(1)) - o.binCuts.begin()); };
    const auto binOf = [&cutsUpTo](uint64_t value) { return uint64_t(ReferencePosition::fromValue(value).getContigId()) << 32 | cutsUpTo(value); };
    std::vector<PairEndIndex> ends[2];
    const bool filtering = o.markDuplicates || !o.keepDuplicates;
    if (filtering || o.realignGaps)
    {
        // BinSorter::loadAlignedData (:214-291) + resolveDuplicates (:293-330): the ends of pairs with a bin position, forward-strand ones apart
        // from reverse-strand ones and shadows; single-ended reads and the unaligned bin are never filtered.  One library (one barcode), and bins
        // as wide as a contig: mates with equal anchors then share a storage bin, so mate_.info_.storageBin_ is the same everywhere
        for (size_t k = 0; k < stored.size(); ++k)
        {
            const Stored &s = stored[k]; const FragmentRecord &h = *s.header;
            if (!s.paired() || h.fStrandPosition == NO_MATCH_VALUE) continue;
            const FragmentRecord &m = *s.mate;
            const unsigned readIndex = (h.flags & 64) ? 1 : 0;
            // getTemplateDuplicateRank (Fragment.hh:66-71): template quality << 32 | (total read length - edit distance) << 16 | template alignment score
            unsigned quality = 0;
            for (unsigned b = 0; b < o.clusterLength; ++b) quality += isBclN(s.clusterBcl[b]) ? 2 : s.clusterBcl[b] >> 2;    // Read.cpp:56-69: an N has quality 2
            const unsigned templateAlignmentScore = (h.reserved >> 16) == 0xffffu ? 0xffffffffu : (h.reserved >> 16);
            PairEndIndex e;
            e.duplicateClusterRank = uint64_t(quality) << 32 | (unsigned(h.readLength) + m.readLength - (unsigned(h.editDistance) + m.editDistance)) << 16 | templateAlignmentScore;
            e.mateAnchor = anchorOf(m, s.clusterBcl + o.readOffset[1 - readIndex]);
            e.mateInfo = unsigned(bool(h.flags & 4)) | unsigned(bool(h.flags & 16)) << 1;
            e.library = 0; e.globalClusterId = uint64_t(h.tile) * CLUSTERS_PER_TILE_FACTOR + h.clusterId; e.tag = k;
            const bool rs = s.reverse() || s.unmapped();
            e.primary = rs ? anchorOf(h, s.clusterBcl + o.readOffset[readIndex]) : h.fStrandPosition;
            ends[rs].push_back(e);
        }
        for (int rs = 1; rs >= 0; --rs)
        {
            // a bin at a time, the bins in position order
            std::map<uint64_t, std::vector<PairEndIndex> > byBin;
            for (const PairEndIndex &e : ends[rs]) byBin[binOf(stored[e.tag].header->fStrandPosition)].push_back(e);
            ends[rs].clear();
            for (auto &bin : byBin)
            {
                std::vector<char> dup;
                filterDuplicates(bin.second, dup);
                if (filtering) for (size_t k = 0; k < bin.second.size(); ++k) if (dup[k]) stored[bin.second[k].tag].duplicate = true;
                ends[rs].insert(ends[rs].end(), bin.second.begin(), bin.second.end());
            }
        }
    }
    if (o.realignGaps)
    {
        // BinSorter::collectGaps (:387-403): the gaps of every fragment of the bin's data (discarded duplicates included), a bin = a contig
        const ContigList &contigs = *o.contigs;
        std::map<uint64_t, RealignerGaps> binGaps;
        for (const Stored &s : stored)
        {
            const FragmentRecord &h = *s.header;
            if (h.fStrandPosition == NO_MATCH_VALUE || (h.flags & 2) || !h.gapCount) continue;
            binGaps[binOf(h.fStrandPosition)].addGaps(ReferencePosition::fromValue(h.fStrandPosition), s.cigarBegin, s.cigarEnd);
        }
        for (auto &g : binGaps) g.second.finalizeGaps();
        // BinSorter::realignGaps (:405-417): the index in order (single-ended, reverse-strand ends and shadows, forward-strand ends), duplicates that were dropped are not in it
        const GapRealigner realigner = { o.realignVigorously, o.realignDodgy, 1, 3, 4, 0, o.clipSemialigned, contigs };      // BinSorter.hh:96-98
        realignedCigars.reserve(size_t(1) << 26);
        std::vector<size_t> order;
        for (size_t k = 0; k < stored.size(); ++k) if (!stored[k].paired() && stored[k].header->fStrandPosition != NO_MATCH_VALUE) order.push_back(k);
        for (int rs = 1; rs >= 0; --rs) for (const PairEndIndex &e : ends[rs]) order.push_back(size_t(e.tag));
        for (const size_t k : order)
        {
            Stored &s = stored[k];
            if (s.duplicate && !o.keepDuplicates) continue;
            FragmentRecord &h = *s.header;
            if (h.flags & 2) continue;
            const ReferencePosition pos = ReferencePosition::fromValue(h.fStrandPosition);
            // the bin: the contig, or the stretch of it between the cuts on either side of the fragment
            ReferencePosition binStartPos(pos.getContigId(), 0), binEndPos(pos.getContigId(), contigs.at(pos.getContigId()).forward.size());
            {
                const uint64_t upTo = cutsUpTo(h.fStrandPosition);
                if (upTo && ReferencePosition::fromValue(o.binCuts[upTo - 1]).getContigId() == pos.getContigId()) binStartPos = ReferencePosition::fromValue(o.binCuts[upTo - 1]);
                if (upTo < o.binCuts.size() && ReferencePosition::fromValue(o.binCuts[upTo]).getContigId() == pos.getContigId()) binEndPos = ReferencePosition::fromValue(o.binCuts[upTo]);
            }
            const uint64_t bin = binOf(h.fStrandPosition);
            RealignFragment f = RealignFragment();
            f.fStrandPosition = pos; f.mateFStrandPosition = h.mateFStrandPosition; f.observedLength = h.observedLength; f.lowClipped = h.lowClipped; f.highClipped = h.highClipped;
            f.alignmentScore = h.alignmentScore; f.templateAlignmentScore = h.templateAlignmentScore; f.readLength = h.readLength; f.editDistance = h.editDistance; f.flags = h.flags; f.bases = s.bases.data();
            RealignIndex index = { pos, s.cigarBegin, s.cigarEnd };
            bool changed = false;
            realigner.realign(binGaps[bin], binStartPos, binEndPos, index, f, realignedCigars, changed);
            if (!changed) continue;
            h.fStrandPosition = f.fStrandPosition.value; h.observedLength = f.observedLength; h.editDistance = f.editDistance;
            s.cigarBegin = index.cigarBegin; s.cigarEnd = index.cigarEnd; h.cigarLength = uint16_t(index.cigarEnd - index.cigarBegin);
            // GapRealigner::updatePairDetails (GapRealigner.cpp:267-318); index.hasMate(): the mate is in the same bin
            const bool hasMate = s.paired() && ReferencePosition::fromValue(h.mateFStrandPosition).getContigId() == pos.getContigId();
            if (!hasMate || (h.flags & 4)) { h.bamTlen = h.bamTlen < 0 ? -int(h.observedLength) + 1 : int(h.observedLength) - 1; continue; }
            FragmentRecord &mate = *s.mate;
            const ReferencePosition fragmentBeginPos = f.fStrandPosition, fragmentEndPos = ReferencePosition::fromValue(advance(h.fStrandPosition, h.observedLength));
            const ReferencePosition mateBeginPos = ReferencePosition::fromValue(h.mateFStrandPosition), mateEndPos = ReferencePosition::fromValue(advance(h.mateFStrandPosition, mate.observedLength));
            // io::FragmentHeader::getTlen (Fragment.hh:199-212)
            const uint64_t distance = std::max(fragmentEndPos, mateEndPos).getLocation() - std::min(fragmentBeginPos, mateBeginPos).getLocation();
            const bool firstRead = h.flags & 32;
            const long tlen = fragmentBeginPos < mateBeginPos ? long(distance) : (mateBeginPos < fragmentBeginPos || !firstRead) ? -long(distance) : long(distance);
            h.bamTlen = int(tlen); mate.bamTlen = -h.bamTlen;
            mate.mateFStrandPosition = h.fStrandPosition;
            // TemplateLengthStatistics::checkModel(fragment, mate) (TemplateLengthStatistics.hh:104-118) on the two FragmentAccessors
            bool proper = false;
            {
                const TemplateLengthStatistics &tls = *s.tls;            // barcodeTemplateLengthStatistics_[fragment.barcode_]
                const ReferencePosition mp = ReferencePosition::fromValue(mate.fStrandPosition);
                if (pos.getContigId() == mp.getContigId())
                {
                    const long p1 = long(f.fStrandPosition.getPosition()), p2 = long(mp.getPosition());
                    const unsigned model = ((p1 <= p2) ? 0 : 4) | ((h.flags & 8) ? 2 : 0) | ((mate.flags & 8) ? 1 : 0);
                    if (model == unsigned(tls.bestModels[0]) || model == unsigned(tls.bestModels[1]))
                    {
                        const unsigned long length = p1 < p2 ? (unsigned long)std::max<long>(p2 + mate.observedLength - p1, h.observedLength) : (unsigned long)std::max<long>(p1 + h.observedLength - p2, mate.observedLength);
                        proper = !(length > tls.max) && !(length < tls.min);
                    }
                }
            }
            h.flags = (h.flags & ~256u) | (proper ? 256u : 0u); mate.flags = (mate.flags & ~256u) | (proper ? 256u : 0u);
        }
    }
    if (!o.keepDuplicates) stored.erase(std::remove_if(stored.begin(), stored.end(), [](const Stored &s) { return s.duplicate; }), stored.end());
    // aligned bins: everything with a bin position; the unaligned bin keeps storage order
    std::vector<const Stored *> aligned, unaligned;
    for (const Stored &s : stored) (s.header->fStrandPosition == NO_MATCH_VALUE ? unaligned : aligned).push_back(&s);
    std::sort(aligned.begin(), aligned.end(), [](const Stored *l, const Stored *r)
    {
        // Index::pos_ is a ReferencePosition; those compare by value (ReferencePosition.hh operator<)
        const uint64_t lp = l->header->fStrandPosition, rp = r->header->fStrandPosition;
        if (lp < rp) return true;
        if (lp == rp)
        {
            const uint64_t lc = l->header->tile * CLUSTERS_PER_TILE_FACTOR + l->header->clusterId, rc = r->header->tile * CLUSTERS_PER_TILE_FACTOR + r->header->clusterId;
            return lc < rc || (lc == rc && (l->unmapped() < r->unmapped() || (l->unmapped() == r->unmapped() && l->secondRead() < r->secondRead())));
        }
        return false;
    });
    // The unaligned bin is written as stored (Build.cpp: no sort for bin 0).  MatchSelector stores tile after tile in the order of the tile
    // indexes (FragmentHeader::tile_ is that index) and, within a tile, cluster after cluster when one thread does the work; with several
    // threads the reference's order inside a tile varies from run to run.  The single-threaded order is the one restated here.
    std::stable_sort(unaligned.begin(), unaligned.end(), [](const Stored *l, const Stored *r)
    {
        return l->header->tile * CLUSTERS_PER_TILE_FACTOR + l->header->clusterId < r->header->tile * CLUSTERS_PER_TILE_FACTOR + r->header->clusterId;
    });
    nRecords = 0;
    for (const Stored *s : aligned) { Adapter a = { *s, s->header->fStrandPosition, true, o }; serializeAlignment(os, a); ++nRecords; }
    unalignedOffset = os.size();
    for (const Stored *s : unaligned) { Adapter a = { *s, NO_MATCH_VALUE, false, o }; serializeAlignment(os, a); ++nRecords; }
}

// bam::serializeHeader (Bam.hh:153-235)
void bamHeader(const std::string &commandLine, const std::string &description, const std::string &version, const std::vector<std::string> &headerLines,
               const std::vector<std::pair<std::string, uint32_t> > &refSeqs, std::vector<char> &os, const std::vector<SqTags> *tags)
{
    std::string text = "@HD\tVN:1.0\tSO:coordinate\n@PG\tID:iSAAC\tPN:iSAAC\tCL:" + commandLine + "\t" + (description.empty() ? std::string() : ("DS:" + description + "\t")) +
                       "VN:" + version + "\n";
    for (const std::string &l : headerLines) text += l + "\n";
    for (size_t i = 0; i < refSeqs.size(); ++i)
    {   // Bam.hh:194-213
        std::string sq = "@SQ\tSN:" + refSeqs[i].first + "\tLN:" + std::to_string(refSeqs[i].second);
        if (tags && (*tags)[i].as.length()) sq += "\tAS:" + (*tags)[i].as;
        if (tags && !(*tags)[i].ur.empty()) sq += "\tUR:" + (*tags)[i].ur;
        if (tags && !(*tags)[i].m5.empty()) sq += "\tM5:" + (*tags)[i].m5;
        text += sq + "\n";
    }
    put(os, "BAM\1", 4); putInt(os, int(text.size())); put(os, text.data(), text.size());
    putInt(os, int(refSeqs.size()));
    for (const auto &r : refSeqs) { putInt(os, int(r.first.size() + 1)); put(os, r.first.c_str(), r.first.size() + 1); putInt(os, int(r.second)); }
}

} // namespace oracle
