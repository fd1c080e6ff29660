// sorted-reference.xml and its mask files: the on-disk form of the k-mer table (reference::SortedReferenceMetadata,
// lib/reference/SortedReferenceXml.cpp:35-330; written by isaac-sort-reference, read by isaac-align -r).  Host code on the public
// C ABI only: the table itself moves through isaac_gpu_load_index / isaac_gpu_get_index_range.
#include "../../include/isaac_gpu.h"
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace
{
thread_local std::string g_xmlError;
int xmlFail(int code, const std::string &what) { g_xmlError = what; return code; }

// ---- a reader for the subset of XML the format uses: elements, attributes, text, <?...?>, <!--...-->, the five entities
struct Node
{
    std::string name, text; std::map<std::string, std::string> attributes; std::vector<std::unique_ptr<Node> > children;
    const Node *child(const char *n) const { for (const auto &c : children) if (c->name == n) return c.get(); return nullptr; }
};
struct Parser
{
    const char *p, *end;
    [[noreturn]] void fail(const std::string &what) const { throw std::runtime_error("sorted-reference.xml: " + what); }
    void skipSpace() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool starts(const char *s) const { const size_t n = std::strlen(s); return size_t(end - p) >= n && 0 == std::memcmp(p, s, n); }
    void skipTo(const char *s) { while (p < end && !starts(s)) ++p; if (p == end) fail(std::string("unterminated ") + s); p += std::strlen(s); }
    static std::string unescape(const std::string &s)
    {
        std::string r; r.reserve(s.size());
        for (size_t i = 0; i < s.size(); ++i)
        {
            if (s[i] != '&') { r += s[i]; continue; }
            static const struct { const char *e; char c; } table[] = { { "&lt;", '<' }, { "&gt;", '>' }, { "&amp;", '&' }, { "&quot;", '"' }, { "&apos;", '\'' } };
            bool done = false;
            for (const auto &t : table) if (0 == s.compare(i, std::strlen(t.e), t.e)) { r += t.c; i += std::strlen(t.e) - 1; done = true; break; }
            if (!done) r += s[i];
        }
        return r;
    }
    std::string name() { const char *b = p; while (p < end && (std::isalnum((unsigned char)*p) || *p == '_' || *p == '-' || *p == ':' || *p == '.')) ++p; if (b == p) fail("element or attribute name expected"); return std::string(b, p); }
    void prolog() { for (;;) { skipSpace(); if (starts("<?")) skipTo("?>"); else if (starts("<!--")) skipTo("-->"); else if (starts("<!")) skipTo(">"); else return; } }
    std::unique_ptr<Node> element()
    {
        if (p == end || *p != '<') fail("'<' expected");
        ++p;
        std::unique_ptr<Node> n(new Node);
        n->name = name();
        for (;;)
        {
            skipSpace();
            if (p == end) fail("unterminated element " + n->name);
            if (*p == '/') { ++p; if (p == end || *p != '>') fail("'>' expected"); ++p; return n; }
            if (*p == '>') { ++p; break; }
            const std::string a = name();
            skipSpace(); if (p == end || *p != '=') fail("'=' expected after attribute " + a); ++p; skipSpace();
            if (p == end || (*p != '"' && *p != '\'')) fail("quoted value expected for attribute " + a);
            const char q = *p++; const char *b = p;
            while (p < end && *p != q) ++p;
            if (p == end) fail("unterminated attribute " + a);
            n->attributes[a] = unescape(std::string(b, p)); ++p;
        }
        for (;;)
        {
            const char *b = p;
            while (p < end && *p != '<') ++p;
            n->text += unescape(std::string(b, p));
            if (p == end) fail("unterminated element " + n->name);
            if (starts("<!--")) { skipTo("-->"); continue; }
            if (starts("<![CDATA[")) { const char *cb = p + 9; skipTo("]]>"); n->text += std::string(cb, p - 3); continue; }
            if (starts("</")) { p += 2; const std::string c = name(); if (c != n->name) fail("</" + c + "> closes <" + n->name + ">"); skipSpace(); if (p == end || *p != '>') fail("'>' expected"); ++p; break; }
            n->children.push_back(element());
        }
        // text of an element with children is layout
        if (!n->children.empty()) n->text.clear();
        else { const size_t b = n->text.find_first_not_of(" \n\r\t"); const size_t e = n->text.find_last_not_of(" \n\r\t"); n->text = b == std::string::npos ? std::string() : n->text.substr(b, e - b + 1); }
        return n;
    }
};

uint64_t number(const Node *n, const char *what)
{
    if (!n) throw std::runtime_error(std::string("sorted-reference.xml: element ") + what + " is missing");
    char *e = nullptr; errno = 0;
    const unsigned long long v = std::strtoull(n->text.c_str(), &e, 10);
    if (errno || e == n->text.c_str() || *e) throw std::runtime_error(std::string("sorted-reference.xml: ") + what + " is not a number: " + n->text);
    return v;
}
uint64_t numberAttribute(const Node &n, const char *a, bool optional, uint64_t fallback)
{
    const auto it = n.attributes.find(a);
    if (it == n.attributes.end()) { if (optional) return fallback; throw std::runtime_error(std::string("sorted-reference.xml: attribute ") + a + " of " + n.name + " is missing"); }
    char *e = nullptr; errno = 0;
    const unsigned long long v = std::strtoull(it->second.c_str(), &e, 10);
    if (errno || e == it->second.c_str() || *e) throw std::runtime_error(std::string("sorted-reference.xml: attribute ") + a + " of " + n.name + " is not a number: " + it->second);
    return v;
}
void copyText(char *dst, size_t cap, const std::string &s, const char *what)
{
    if (s.size() >= cap) throw std::runtime_error(std::string("sorted-reference.xml: ") + what + " is too long");
    std::memset(dst, 0, cap); std::memcpy(dst, s.data(), s.size());
}
std::string escape(const char *s)
{
    std::string r;
    for (; *s; ++s) switch (*s) { case '<': r += "&lt;"; break; case '>': r += "&gt;"; break; case '&': r += "&amp;"; break; case '"': r += "&quot;"; break; default: r += *s; }
    return r;
}
const unsigned CURRENT_REFERENCE_FORMAT_VERSION = 3, OLDEST_SUPPORTED_REFERENCE_FORMAT_VERSION = 2;     // include/reference/SortedReferenceMetadata.hh:38-39
} // namespace

extern "C" {

const char *isaac_gpu_sorted_reference_last_error(void) { return g_xmlError.c_str(); }

// loadSortedReferenceXml (SortedReferenceXml.cpp:35-213)
int isaac_gpu_sorted_reference_parse(const char *xml, uint64_t nBytes, isaac_reference_contig *contigs, uint32_t contigCapacity, uint32_t *nContigs,
                                     isaac_reference_mask_file *masks, uint32_t maskCapacity, uint32_t *nMasks, uint32_t *formatVersion)
{
    try
    {
        if (!xml || !nContigs || !nMasks) return xmlFail(ISAAC_GPU_EINVAL, "null argument");
        Parser parser; parser.p = xml; parser.end = xml + nBytes;
        parser.prolog();
        const std::unique_ptr<Node> root = parser.element();
        if (root->name != "SortedReference") return xmlFail(ISAAC_GPU_EFORMAT, "sorted-reference.xml: the document element is not SortedReference");
        const uint64_t version = number(root->child("FormatVersion"), "FormatVersion");
        if (version > CURRENT_REFERENCE_FORMAT_VERSION || version < OLDEST_SUPPORTED_REFERENCE_FORMAT_VERSION)
            return xmlFail(ISAAC_GPU_EFORMAT, "Unexpected sorted reference FormatVersion: " + std::to_string(version) + ". FormatVersion must be in range [" +
                           std::to_string(OLDEST_SUPPORTED_REFERENCE_FORMAT_VERSION) + "," + std::to_string(CURRENT_REFERENCE_FORMAT_VERSION) + "]");
        if (formatVersion) *formatVersion = CURRENT_REFERENCE_FORMAT_VERSION;       // a successfully read file is bumped to the current version (:188-190)
        uint32_t nc = 0, nm = 0;
        if (const Node *list = root->child("Contigs"))
            for (const auto &cp : list->children)
            {
                if (cp->name != "Contig") continue;
                if (nc < contigCapacity && contigs)
                {
                    isaac_reference_contig &c = contigs[nc];
                    std::memset(&c, 0, sizeof(c));
                    c.genomic_position = numberAttribute(*cp, "Position", false, 0);
                    c.index = uint32_t(number(cp->child("Index"), "Index"));
                    c.karyotype_index = cp->child("KaryotypeIndex") ? uint32_t(number(cp->child("KaryotypeIndex"), "KaryotypeIndex")) : c.index;    // absent in old files (:97-105)
                    if (!cp->child("Name")) throw std::runtime_error("sorted-reference.xml: element Name is missing");
                    copyText(c.name, sizeof(c.name), cp->child("Name")->text, "Name");
                    const Node *sequence = cp->child("Sequence");
                    if (!sequence || !sequence->child("File")) throw std::runtime_error("sorted-reference.xml: element Sequence/File is missing");
                    copyText(c.file, sizeof(c.file), sequence->child("File")->text, "Sequence/File");
                    c.offset = number(sequence->child("Offset"), "Offset"); c.size = number(sequence->child("Size"), "Size");
                    c.total_bases = number(cp->child("TotalBases"), "TotalBases"); c.acgt_bases = number(cp->child("AcgtBases"), "AcgtBases");
                    if (const Node *bam = cp->child("BamMetadata")) if (const Node *sq = bam->child("Sq"))
                    {
                        if (sq->child("As")) copyText(c.bam_sq_as, sizeof(c.bam_sq_as), sq->child("As")->text, "As");
                        if (sq->child("Ur")) copyText(c.bam_sq_ur, sizeof(c.bam_sq_ur), sq->child("Ur")->text, "Ur");
                        if (sq->child("M5")) copyText(c.bam_m5, sizeof(c.bam_m5), sq->child("M5")->text, "M5");
                    }
                }
                ++nc;
            }
        if (const Node *permutations = root->child("Permutations"))
        {
            const Node *permutation = permutations->child("Permutation");
            if (!permutation) return xmlFail(ISAAC_GPU_EFORMAT, "sorted-reference.xml: element Permutation is missing");
            const auto name = permutation->attributes.find("Name");
            if (name == permutation->attributes.end() || name->second != "ABCD") return xmlFail(ISAAC_GPU_EFORMAT, "Only ABCD permutation masks are supported");
            std::map<uint64_t, bool> seen;
            for (const auto &mp : permutation->children)
            {
                if (mp->name != "Masks") continue;
                const uint64_t width = numberAttribute(*mp, "Width", false, 0), seedLength = numberAttribute(*mp, "SeedLength", true, 32);     // absent SeedLength = 32 (:68-70)
                if (seen[seedLength]) return xmlFail(ISAAC_GPU_EFORMAT, "Multiple Masks elements with same SeedLength are not allowed");
                seen[seedLength] = true;
                for (const auto &fp : mp->children)
                {
                    if (fp->name != "Mask") continue;
                    if (nm < maskCapacity && masks)
                    {
                        isaac_reference_mask_file &m = masks[nm];
                        std::memset(&m, 0, sizeof(m));
                        m.mask_width = uint32_t(width); m.seed_length = uint32_t(seedLength); m.mask = uint32_t(numberAttribute(*fp, "Mask", false, 0));
                        if (!fp->child("File")) throw std::runtime_error("sorted-reference.xml: element Mask/File is missing");
                        copyText(m.file, sizeof(m.file), fp->child("File")->text, "Mask/File");
                        const Node *kmers = fp->child("Kmers");
                        m.kmers = number(kmers ? kmers->child("Total") : nullptr, "Kmers/Total");
                    }
                    ++nm;
                }
            }
        }
        *nContigs = nc; *nMasks = nm;
        if ((contigs && nc > contigCapacity) || (masks && nm > maskCapacity)) return xmlFail(ISAAC_GPU_ECAPACITY, "more contigs or mask files than the caller has room for");
        return 0;
    }
    catch (const std::exception &e) { return xmlFail(ISAAC_GPU_EFORMAT, e.what()); }
}

// saveSortedReferenceXml (SortedReferenceXml.cpp:216-330)
int isaac_gpu_sorted_reference_format(const isaac_reference_contig *contigs, uint32_t nContigs, const isaac_reference_mask_file *masks, uint32_t nMasks,
                                      char *xmlOut, uint64_t capacity, uint64_t *nBytesOut)
{
    std::string s = "<?xml version=\"1.0\"?>\n<SortedReference>\n";
    s += "  <FormatVersion>" + std::to_string(CURRENT_REFERENCE_FORMAT_VERSION) + "</FormatVersion>\n";
    s += "  <SoftwareVersion>iSAAC-01.15.04.01 sorted-reference format, written by isaac_aligner_amd</SoftwareVersion>\n";
    if (nContigs)
    {
        s += "  <Contigs>\n";
        for (uint32_t i = 0; i < nContigs; ++i)
        {
            const isaac_reference_contig &c = contigs[i];
            s += "    <Contig Position=\"" + std::to_string(c.genomic_position) + "\">\n";
            s += "      <Index>" + std::to_string(c.index) + "</Index>\n      <KaryotypeIndex>" + std::to_string(c.karyotype_index) + "</KaryotypeIndex>\n";
            s += "      <Name>" + escape(c.name) + "</Name>\n";
            s += "      <Sequence>\n        <File>" + escape(c.file) + "</File>\n        <Offset>" + std::to_string(c.offset) + "</Offset>\n        <Size>" + std::to_string(c.size) + "</Size>\n      </Sequence>\n";
            s += "      <TotalBases>" + std::to_string(c.total_bases) + "</TotalBases>\n      <AcgtBases>" + std::to_string(c.acgt_bases) + "</AcgtBases>\n";
            s += "      <BamMetadata>\n        <Sq>\n          <As>" + escape(c.bam_sq_as) + "</As>\n          <Ur>" + escape(c.bam_sq_ur) + "</Ur>\n          <M5>" + escape(c.bam_m5) + "</M5>\n        </Sq>\n      </BamMetadata>\n";
            s += "    </Contig>\n";
        }
        s += "  </Contigs>\n";
    }
    if (nMasks)
    {
        s += "  <Permutations>\n    <Permutation Name=\"ABCD\">\n";
        for (uint32_t i = 0; i < nMasks;)
        {   // one Masks element per seed length, in the order given
            uint32_t j = i;
            s += "      <Masks Width=\"" + std::to_string(masks[i].mask_width) + "\" SeedLength=\"" + std::to_string(masks[i].seed_length) + "\">\n";
            for (; j < nMasks && masks[j].seed_length == masks[i].seed_length; ++j)
                s += "        <Mask Mask=\"" + std::to_string(masks[j].mask) + "\">\n          <File>" + escape(masks[j].file) + "</File>\n          <Kmers>\n            <Total>" + std::to_string(masks[j].kmers) +
                     "</Total>\n          </Kmers>\n        </Mask>\n";
            s += "      </Masks>\n";
            i = j;
        }
        s += "    </Permutation>\n  </Permutations>\n";
    }
    s += "</SortedReference>\n";
    if (nBytesOut) *nBytesOut = s.size();
    if (!xmlOut) return 0;
    if (s.size() + 1 > capacity) return xmlFail(ISAAC_GPU_ECAPACITY, "xml_out is too small");
    std::memcpy(xmlOut, s.c_str(), s.size() + 1);
    return 0;
}

namespace { struct Mapping { void *p = MAP_FAILED; size_t n = 0; ~Mapping() { if (p != MAP_FAILED) munmap(p, n); } }; }

// what isaac-align does with -r: the XML's 32-mer mask files mapped and handed to the context in mask order, the contig
// translation taken from <Index> / <KaryotypeIndex> (MatchFinder.cpp:51-66).  Relative <File> paths: relative to the XML.
int isaac_gpu_load_sorted_reference(isaac_gpu_ctx *ctx, const char *xmlPath)
{
    try
    {
        if (!ctx || !xmlPath) return xmlFail(ISAAC_GPU_EINVAL, "null argument");
        FILE *f = std::fopen(xmlPath, "rb");
        if (!f) return xmlFail(ISAAC_GPU_EINVAL, std::string("Failed to open sorted reference file ") + xmlPath + ": " + std::strerror(errno));
        std::string text; char buffer[65536]; size_t got;
        while ((got = std::fread(buffer, 1, sizeof(buffer), f)) > 0) text.append(buffer, got);
        std::fclose(f);
        uint32_t nContigs = 0, nMasks = 0, version = 0;
        int rc = isaac_gpu_sorted_reference_parse(text.data(), text.size(), nullptr, 0, &nContigs, nullptr, 0, &nMasks, &version);
        if (rc) return rc;
        std::vector<isaac_reference_contig> contigs(nContigs); std::vector<isaac_reference_mask_file> masks(nMasks);
        rc = isaac_gpu_sorted_reference_parse(text.data(), text.size(), contigs.data(), nContigs, &nContigs, masks.data(), nMasks, &nMasks, &version);
        if (rc) return rc;
        std::vector<isaac_reference_mask_file> wanted;
        for (const auto &m : masks) if (32 == m.seed_length) wanted.push_back(m);
        if (wanted.empty()) return xmlFail(ISAAC_GPU_EINVAL, "the sorted reference has no mask files for seed length 32");
        std::sort(wanted.begin(), wanted.end(), [](const isaac_reference_mask_file &a, const isaac_reference_mask_file &b) { return a.mask < b.mask; });
        std::string directory(xmlPath);
        const size_t slash = directory.find_last_of('/');
        directory = slash == std::string::npos ? std::string(".") : directory.substr(0, slash);
        std::vector<Mapping> mappings(wanted.size());
        std::vector<const isaac_reference_kmer *> pointers(wanted.size()); std::vector<uint64_t> sizes(wanted.size());
        for (size_t i = 0; i < wanted.size(); ++i)
        {
            const std::string path = wanted[i].file[0] == '/' ? std::string(wanted[i].file) : directory + "/" + wanted[i].file;
            const int fd = open(path.c_str(), O_RDONLY);
            if (fd < 0) return xmlFail(ISAAC_GPU_EINVAL, "Failed to open sorted reference file " + path + ": " + std::strerror(errno));
            struct stat st;
            if (fstat(fd, &st)) { close(fd); return xmlFail(ISAAC_GPU_EINVAL, "Failed to stat " + path); }
            if (uint64_t(st.st_size) != wanted[i].kmers * sizeof(isaac_reference_kmer))
            { close(fd); return xmlFail(ISAAC_GPU_EFORMAT, path + " holds " + std::to_string(st.st_size / sizeof(isaac_reference_kmer)) + " records, the metadata says " + std::to_string(wanted[i].kmers)); }
            if (st.st_size)
            {
                mappings[i].n = size_t(st.st_size);
                mappings[i].p = mmap(nullptr, mappings[i].n, PROT_READ, MAP_PRIVATE, fd, 0);
                if (MAP_FAILED == mappings[i].p) { close(fd); return xmlFail(ISAAC_GPU_ENOMEM, "Failed to map " + path + ": " + std::strerror(errno)); }
            }
            close(fd);
            pointers[i] = MAP_FAILED == mappings[i].p ? nullptr : static_cast<const isaac_reference_kmer *>(mappings[i].p); sizes[i] = wanted[i].kmers;
        }
        // <Index> and <KaryotypeIndex> must each name every contig once: a duplicate would leave another contig translated to 0
        std::vector<uint32_t> karyotype(nContigs);
        std::vector<char> seenIndex(nContigs, 0), seenKaryotype(nContigs, 0);
        for (const auto &c : contigs)
        {
            if (c.index >= nContigs || c.karyotype_index >= nContigs) return xmlFail(ISAAC_GPU_EFORMAT, "contig Index / KaryotypeIndex out of range");
            if (seenIndex[c.index]++ || seenKaryotype[c.karyotype_index]++)
                return xmlFail(ISAAC_GPU_EFORMAT, std::string("contig Index / KaryotypeIndex values are not a permutation of 0..n-1 (contig ") + c.name + ")");
            karyotype[c.index] = c.karyotype_index;
        }
        rc = isaac_gpu_load_index(ctx, pointers.data(), sizes.data(), uint32_t(wanted.size()), nContigs ? karyotype.data() : nullptr, nContigs);
        if (rc) return xmlFail(rc, isaac_gpu_last_error());
        return 0;
    }
    catch (const std::exception &e) { return xmlFail(ISAAC_GPU_EHIP, e.what()); }
}

// what isaac-sort-reference leaves behind: <directory>/<genome name>-32mer-6bit-ABCD-NN.dat for the 64 masks of the resident table
// and <directory>/sorted-reference.xml describing them and the contigs (contigs[i].index / karyotype_index / name ... as the caller
// knows them; total_bases etc. are written as given)
int isaac_gpu_save_sorted_reference(isaac_gpu_ctx *ctx, const char *directory, const char *genomeName, const isaac_reference_contig *contigs, uint32_t nContigs)
{
    try
    {
        if (!ctx || !directory || !genomeName) return xmlFail(ISAAC_GPU_EINVAL, "null argument");
        const uint32_t nMasks = 64;
        std::vector<uint64_t> offsets(nMasks + 1);
        int rc = isaac_gpu_get_mask_offsets(ctx, offsets.data(), nMasks);
        if (rc) return xmlFail(rc, isaac_gpu_last_error());
        std::vector<isaac_reference_mask_file> masks(nMasks);
        std::vector<isaac_reference_kmer> records;
        for (uint32_t m = 0; m < nMasks; ++m)
        {
            const uint64_t n = offsets[m + 1] - offsets[m];
            char name[64]; std::snprintf(name, sizeof(name), "-32mer-6bit-ABCD-%02u.dat", m);
            const std::string file = std::string(genomeName) + name, path = std::string(directory) + "/" + file;
            records.resize(n);
            if (n) { rc = isaac_gpu_get_index_range(ctx, offsets[m], n, records.data()); if (rc) return xmlFail(rc, isaac_gpu_last_error()); }
            FILE *f = std::fopen(path.c_str(), "wb");
            if (!f) return xmlFail(ISAAC_GPU_EINVAL, "Failed to open " + path + " for writing: " + std::strerror(errno));
            const size_t written = n ? std::fwrite(records.data(), sizeof(isaac_reference_kmer), n, f) : 0;
            if (std::fclose(f) || written != n) return xmlFail(ISAAC_GPU_EINVAL, "Failed to write " + path);
            std::memset(&masks[m], 0, sizeof(masks[m]));
            masks[m].mask_width = 6; masks[m].mask = m; masks[m].seed_length = 32; masks[m].kmers = n;
            copyText(masks[m].file, sizeof(masks[m].file), file, "mask file name");
        }
        uint64_t bytes = 0;
        isaac_gpu_sorted_reference_format(contigs, nContigs, masks.data(), nMasks, nullptr, 0, &bytes);
        std::string xml(bytes + 1, '\0');
        rc = isaac_gpu_sorted_reference_format(contigs, nContigs, masks.data(), nMasks, &xml[0], xml.size(), &bytes);
        if (rc) return rc;
        const std::string path = std::string(directory) + "/sorted-reference.xml";
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return xmlFail(ISAAC_GPU_EINVAL, "Failed to open sorted reference file for write: " + path);
        const size_t written = std::fwrite(xml.data(), 1, bytes, f);
        if (std::fclose(f) || written != bytes) return xmlFail(ISAAC_GPU_EINVAL, "Failed to write " + path);
        return 0;
    }
    catch (const std::exception &e) { return xmlFail(ISAAC_GPU_EHIP, e.what()); }
}

} // extern "C"
