#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference's own cppunit known-answer tests.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What is extracted is DATA: the literal inputs and the asserted outputs of
  * lib/alignment/cppunit/testSimpleIndelAligner.cpp:264-615   -> simple_indel.json
  * lib/alignment/cppunit/testFragmentBuilder2.cpp:183-307     -> fragment_builder2.json
  * lib/alignment/cppunit/testSequencingAdapter.cpp:41-515 (+ the presets of lib/flowcell/SequencingAdapterMetadata.cpp:29-39) -> sequencing_adapter.json
  * lib/alignment/cppunit/testBandedSmithWaterman.cpp:79-225   -> bsw.json (the construction recipe of the test is re-run here
    on genomes drawn with glibc rand() exactly like getGenome() at :33-43; the asserted CIGARs are the test's literals)
  * lib/alignment/cppunit/testSeedId.cpp                        -> seed_id.json
  * lib/alignment/cppunit/testTemplateLengthStatistics.cpp:42-367 -> template_length_statistics.json (asserted literals only)
  * lib/alignment/cppunit/testSemialignedClipper.cpp:189-251, testOverlappingEndsClipper.cpp:109-157 -> clippers.json
  * lib/alignment/cppunit/testTemplateBuilder.cpp:96-373 (+ BuilderInit.hh fixture recipe) -> template_builder.json
  * lib/alignment/cppunit/testShadowAligner.cpp:56-252 -> shadow_aligner.json
  * lib/alignment/cppunit/testFragmentBuilder.cpp:33-598 (seed matches -> candidates) -> fragment_builder.json
  * lib/build/cppunit/testDuplicateFiltering.cpp:131-399 -> duplicate_filtering.json
  * lib/build/cppunit/testGapRealigner.cpp:53-1733 -> gap_realigner.json (the fixture recipe re-run on the test's strings; asserted literals)
  * testMatchFinderClusterInfo.cpp, oligo/cppunit/testKmerGenerator.cpp, oligo/cppunit/testPermutate.cpp,
    reference/cppunit/testNeighborsFinder.cpp -> oligo.json
No reference source text is stored.
"""
import ctypes
import json
import os
import re

REF = "/root/reference/src/c++/lib/alignment/cppunit"
OUT = os.path.dirname(os.path.abspath(__file__))


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    return s


def split_args(argtext):
    """split a C++ argument list at top-level commas; adjacent string literals are concatenated"""
    args, cur, depth, i, in_str = [], "", 0, 0, False
    while i < len(argtext):
        c = argtext[i]
        if in_str:
            cur += c
            if c == "\\":
                cur += argtext[i + 1]
                i += 1
            elif c == '"':
                in_str = False
        else:
            if c == '"':
                in_str = True
                cur += c
            elif c in "([{":
                depth += 1
                cur += c
            elif c in ")]}":
                depth -= 1
                cur += c
            elif c == "," and depth == 0:
                args.append(cur.strip())
                cur = ""
            else:
                cur += c
        i += 1
    if cur.strip():
        args.append(cur.strip())
    out = []
    for a in args:
        lits = re.findall(r'"((?:[^"\\]|\\.)*)"', a)
        if lits and re.fullmatch(r'(\s*"(?:[^"\\]|\\.)*"\s*)+', a):
            out.append(("str", "".join(lits)))
        else:
            out.append(("id", a))
    return out


def find_call(text, start):
    """text[start] is just after 'align(' ; returns (argtext, end index after ')')"""
    depth, i, in_str = 1, start, False
    while depth:
        c = text[i]
        if in_str:
            if c == "\\":
                i += 1
            elif c == '"':
                in_str = False
        else:
            if c == '"':
                in_str = True
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
        i += 1
    return text[start:i - 1], i


def enclosing_block(text, pos):
    depth, i = 0, pos
    while i >= 0:
        if text[i] == "}":
            depth += 1
        elif text[i] == "{":
            if depth == 0:
                break
            depth -= 1
        i -= 1
    begin = i
    depth, j = 0, pos
    while j < len(text):
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            if depth == 0:
                break
            depth -= 1
        j += 1
    return begin, j


def make_simple_indel():
    text = strip_comments(open(os.path.join(REF, "testSimpleIndelAligner.cpp")).read())
    body = text[text.index("void TestSimpleIndelAligner::testEverything()"):]
    cases = []
    for m in re.finditer(r"\balign\(", body):
        argtext, end = find_call(body, m.end())
        args = split_args(argtext)
        b, e = enclosing_block(body, m.start())
        before, after = body[b:m.start()], body[end:e]
        case = {"read": args[0][1], "reference": args[1][1], "seeds": None, "left_clip0": 0, "right_clip1": 0, "expect": {}}
        if len(args) == 4:
            seeds = re.findall(r"SeedMetadata\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\)", before)
            assert len(seeds) == 2 and all(s[1] == "32" and s[2] == "0" for s in seeds) and [s[3] for s in seeds] == ["0", "1"], seeds
            case["seeds"] = [int(seeds[0][0]), int(seeds[1][0])]
        lc = re.search(r"fragmentMetadataList\[0\]\.leftClipped\(\)\s*=\s*(\d+)", before)
        rc = re.search(r"fragmentMetadataList\[1\]\.rightClipped\(\)\s*=\s*(\d+)", before)
        if lc:
            case["left_clip0"] = int(lc.group(1))
        if rc:
            case["right_clip1"] = int(rc.group(1))
        for am in re.finditer(r"CPPUNIT_ASSERT_EQUAL\((.*?),\s*(?:unsigned\()?fragmentMetadataList\[0\]\.(\w+)\(\)(?:\.getPosition\(\))?\)?\);", after):
            expected, what = am.group(1).strip(), am.group(2)
            sm = re.match(r'std::string\("(.*)"\)', expected)
            case["expect"][what] = sm.group(1) if sm else int(re.match(r"(\d+)", expected).group(1))
        assert "getCigarString" in case["expect"], after
        cases.append(case)
    assert len(cases) == 22, len(cases)
    json.dump({"source": "lib/alignment/cppunit/testSimpleIndelAligner.cpp:264-615", "scores": [0, -1, -2, -1, -5], "gap_limit": 20000, "cases": cases},
              open(os.path.join(OUT, "simple_indel.json"), "w"), indent=1)
    return len(cases)


def make_fragment_builder2():
    text = strip_comments(open(os.path.join(REF, "testFragmentBuilder2.cpp")).read())
    cases = []
    for fm in re.finditer(r"void TestFragmentBuilder2::(test\w+)\(\)\s*\{", text):
        name = fm.group(1)
        if name == "testEverything":
            continue
        b = fm.end()
        _, e = enclosing_block(text, b)
        body = text[b:e]
        m = re.search(r"\balign\(", body)
        argtext, end = find_call(body, m.end())
        args = split_args(argtext)
        case = {"name": name, "read": args[0][1], "reference": args[1][1], "reverse": "reverse = true" in body[:m.start()],
                "position": None, "gapped": len(args) == 5 and args[4][1] == "true", "expect": {}}
        pm = re.search(r"fragmentMetadata\.position\s*=\s*(-?\d+)", body[:m.start()])
        if pm:
            case["position"] = int(pm.group(1))
        after = body[end:]
        for am in re.finditer(r"CPPUNIT_ASSERT_EQUAL\((.*?),\s*(?:unsigned\(\*)?fragmentMetadata\.(\w+)\(\)\)?\);", after):
            expected, what = am.group(1).strip(), am.group(2)
            sm = re.match(r'std::string\("(.*)"\)', expected)
            rp = re.match(r"isaac::reference::ReferencePosition\((\d+),\s*(\d+)U?\)", expected)
            case["expect"][what] = sm.group(1) if sm else [int(rp.group(1)), int(rp.group(2))] if rp else int(re.match(r"(\d+)", expected).group(1))
        cases.append(case)
    assert len(cases) == 5, len(cases)
    json.dump({"source": "lib/alignment/cppunit/testFragmentBuilder2.cpp:183-307", "scores": [2, -1, -15, -3, 25], "cases": cases},
              open(os.path.join(OUT, "fragment_builder2.json"), "w"), indent=1)
    return len(cases)


def make_sequencing_adapter():
    """testSequencingAdapter.cpp: the adapter pairs the suite constructs, and for each test that testEverything() runs its read, reference, strand, adapter list
    and every value it asserts; plus the three --default-adapters presets (flowcell/SequencingAdapterMetadata.cpp) as data"""
    text = strip_comments(open(os.path.join(REF, "testSequencingAdapter.cpp")).read())
    metadata = {}
    for m in re.finditer(r"SequencingAdapterMetadata\s+(\w+)\(\s*\"([ACGT]+)\"\s*,\s*(true|false)\s*,\s*(strlen\(\"([ACGT]+)\"\)|\d+)\s*\)", text):
        name, sequence, reverse, clip = m.group(1), m.group(2), m.group(3) == "true", m.group(4)
        metadata[name] = {"sequence": sequence, "reverse": reverse, "clip_length": len(m.group(5)) if clip.startswith("strlen") else int(clip)}
    lists = {}
    for m in re.finditer(r"(\w+)\s*=\s*boost::assign::list_of\(isaac::alignment::matchSelector::SequencingAdapter\((\w+)\)\)\s*\(isaac::alignment::matchSelector::SequencingAdapter\((\w+)\)\)", text):
        lists[m.group(1)] = [metadata[m.group(2)], metadata[m.group(3)]]
    assert sorted(lists) == ["matePairAdapters", "standardAdapters"], lists
    everything = text[text.index("void TestSequencingAdapter::testEverything()"):]
    everything = everything[:everything.index("}")]
    run = re.findall(r"\b(test\w+)\(\);", everything)
    cases = []
    for fm in re.finditer(r"void TestSequencingAdapter::(test\w+)\(\)\s*\{", text):
        name = fm.group(1)
        if name == "testEverything":
            continue
        b = fm.end()
        _, e = enclosing_block(text, b)
        body = text[b:e]
        m = re.search(r"\balign\(", body)
        argtext, end = find_call(body, m.end())
        args = split_args(argtext)
        assert len(args) == 4 and args[2][1] in lists, args
        case = {"name": name, "read": args[0][1], "reference": args[1][1], "adapters": args[2][1], "reverse": "reverse = true" in body[:m.start()], "expect": {}}
        for am in re.finditer(r"CPPUNIT_ASSERT_EQUAL\((.*?),\s*fragmentMetadata\.(\w+)\(\)\);", body[end:]):
            expected, what = am.group(1).strip(), am.group(2)
            sm = re.match(r'std::string\("(.*)"\)', expected)
            rp = re.match(r"isaac::reference::ReferencePosition\((\d+),\s*(\d+)U?\)", expected)
            case["expect"][what] = sm.group(1) if sm else [int(rp.group(1)), int(rp.group(2))] if rp else int(re.match(r"(\d+)", expected).group(1))
        assert "getCigarString" in case["expect"], body
        cases.append(case)
    assert sorted(c["name"] for c in cases) == sorted(run) and len(cases) == 15, (len(cases), run)
    presets_text = strip_comments(open("/root/reference/src/c++/lib/flowcell/SequencingAdapterMetadata.cpp").read())
    presets = {}
    for m in re.finditer(r"SequencingAdapterMetadataList\s+(\w+)\s*=(.*?);", presets_text, flags=re.S):
        entries = []
        for a in re.finditer(r"SequencingAdapterMetadata\(\"([ACGT]+)\"\s*,\s*(true|false)\s*(?:,\s*(\d+))?\)", m.group(2)):
            entries.append({"sequence": a.group(1), "reverse": a.group(2) == "true", "clip_length": len(a.group(1)) if a.group(3) is None else int(a.group(3))})
        presets[m.group(1)] = entries
    assert sorted(presets) == ["NEXTERA_MATEPAIR_ADAPTERS", "NEXTERA_STANDARD_ADAPTERS", "STANDARD_ADAPTERS"], presets
    json.dump({"source": "lib/alignment/cppunit/testSequencingAdapter.cpp:41-515, lib/flowcell/SequencingAdapterMetadata.cpp:29-39", "scores": [2, -1, -15, -3, 25],
               "adapter_lists": lists, "presets": presets, "cases": cases}, open(os.path.join(OUT, "sequencing_adapter.json"), "w"), indent=1)
    return len(cases)


def make_bsw():
    """testBandedSmithWaterman.cpp: the genome is 1000 draws of "ACGT"[rand() % 4].  The rand() state at the time the fixture
    is constructed depends on how many draws other cppunit fixtures made before it, so the recipe is evaluated on the first
    eight consecutive 1000-base genomes of the glibc rand() stream (default seed 1); the asserted CIGARs hold for any of them
    (the test is written to be genome independent: "no Ts to ensure constant location")."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    genomes = ["".join("ACGT"[libc.rand() % 4] for _ in range(1000)) for _ in range(8)]
    cases = []
    for gi, genome in enumerate(genomes):
        # testUngapped :79-103
        database = genome[100:215]
        for i in range(0, 16):
            cases.append({"name": "ungapped", "genome": gi, "query": database[i:i + 100], "database": database, "cigar": [1600]})
        # testSingleDeletion :105-131
        left = right = 40
        deletion = "AGAGCAGCGAGCGACAGCAGCAGCAAA"
        for dlen in range(1, 14):
            dl = 7 - (dlen // 2)
            dlS = genome[100:100 + dl]
            leftS = genome[100 + dl:100 + dl + left - 1] + "T"
            rightS = genome[100 + dl + left:100 + dl + left + right]
            delS = deletion[:dlen]
            drD = 15 - dl - len(delS)
            drDS = genome[100 + dl + left + right:100 + dl + left + right + drD]
            cases.append({"name": "single_deletion", "genome": gi, "query": leftS + rightS, "database": dlS + leftS + delS + rightS + drDS,
                          "cigar": [(left << 4) | 0, (len(delS) << 4) | 2, (right << 4) | 0]})
        # testSingleInsertion :133-154
        database = genome[100:320]
        qlen = len(database) - 15
        for ilen in range(1, 10):
            if database[9 + 100 - 1] == "T":
                # the inserted bases are Ts: when the base in front of the insertion point is a T as well, "99M<n>I" scores
                # the same as the asserted "100M<n>I" and the recipe has no unique answer -> not a known-answer vector
                continue
            left = 100
            right = qlen - left - ilen
            dl = 9
            query = database[dl:dl + left] + "T" * ilen + database[left + dl:left + dl + right]
            cases.append({"name": "single_insertion", "genome": gi, "query": query, "database": database,
                          "cigar": [(left << 4) | 0, (ilen << 4) | 1, (right << 4) | 0]})
        # testMultipleIndels :156-212
        left = center = right = 20
        dl = 6
        dlS = genome[100:100 + dl]
        leftS = genome[100 + dl:100 + dl + left - 1] + "T"
        ins1, ins2 = "A", "CG"
        centerS = genome[100 + dl + left:100 + dl + left + center - 1] + "T"
        del1, del2 = "AAG", "ACAG"
        rightS = genome[100 + dl + left + center:100 + dl + left + center + right]
        tail = 100 + dl + left + center + right
        drID = 15 - dl + len(ins1) - len(del2)
        cases.append({"name": "ins_del", "genome": gi, "query": leftS + ins1 + centerS + rightS, "database": dlS + leftS + centerS + del2 + rightS + genome[tail:tail + drID],
                      "cigar": [(left << 4) | 0, (len(ins1) << 4) | 1, (center << 4) | 0, (len(del2) << 4) | 2, (right << 4) | 0]})
        drI2 = 15 - dl + len(ins1) + len(ins2)
        cases.append({"name": "ins_ins", "genome": gi, "query": leftS + ins1 + centerS + ins2 + rightS, "database": dlS + leftS + centerS + rightS + genome[tail:tail + drI2],
                      "cigar": [(left << 4) | 0, (len(ins1) << 4) | 1, (center << 4) | 0, (len(ins2) << 4) | 1, (right << 4) | 0]})
        drD2 = 15 - dl - len(del1) - len(del2)
        cases.append({"name": "del_del", "genome": gi, "query": leftS + centerS + rightS, "database": dlS + leftS + del1 + centerS + del2 + rightS + genome[tail:tail + drD2],
                      "cigar": [(left << 4) | 0, (len(del1) << 4) | 2, (center << 4) | 0, (len(del2) << 4) | 2, (right << 4) | 0]})
    for c in cases:
        assert len(c["database"]) == len(c["query"]) + 15, c["name"]
    overflow = [  # testOverflow :214-225: (match, mismatch, open, extend, maxReadLength, throws)
        [2, -1, 6, 3, 5460, False], [2, -1, 7, 3, 4681, True], [2, -1, 17, 3, 3681, True], [2, -1, 11, 3, 13681, True]]
    json.dump({"source": "lib/alignment/cppunit/testBandedSmithWaterman.cpp:45-225", "scores": [2, -1, 15, 3], "max_read_length": 300,
               "cases": cases, "overflow": overflow}, open(os.path.join(OUT, "bsw.json"), "w"))
    return len(cases)


def make_seed_id():
    """testSeedId.cpp:35-122: the test is written against the symbolic masks; the data below are those masks spelled out
    (widths tile 12 / barcode 12 / cluster 31 / seed 8 / reverse 1, SeedId.hh:64-68) plus the literal `other` case."""
    masks = {"tile": (1 << 12) - 1, "barcode": (1 << 12) - 1, "cluster": (1 << 31) - 1, "seed": (1 << 8) - 1, "reverse": 1}
    order = ["tile", "barcode", "cluster", "seed", "reverse"]
    valid = [[0, 0, 0, 0, 0], [masks[k] for k in order], [4020, 1234, 1234567, 3, 1]]
    for i, k in enumerate(order):
        v = [0] * 5
        v[i] = masks[k]
        valid.append(v)
    throws = []
    for i, k in enumerate(order):
        v = [0] * 5
        v[i] = masks[k] + 1
        throws.append(v)
    json.dump({"source": "lib/alignment/cppunit/testSeedId.cpp:35-122", "order": order, "valid": valid, "throws": throws},
              open(os.path.join(OUT, "seed_id.json"), "w"), indent=1)
    return len(valid), len(throws)


def make_template_length_statistics():
    """every CPPUNIT_ASSERT_EQUAL of testTemplateLengthStatistics.cpp as data: alignment models / classes by name, mate orientation and
    mate position windows for the eight model pairs, and the statistics the literal addTemplates() sequence must produce"""
    text = strip_comments(open(os.path.join(REF, "testTemplateLengthStatistics.cpp")).read())
    models = {"FFp": 0, "FRp": 1, "RFp": 2, "RRp": 3, "FFm": 4, "FRm": 5, "RFm": 6, "RRm": 7}
    out = {"source": "lib/alignment/cppunit/testTemplateLengthStatistics.cpp:42-367", "models": models}
    # testAlignmentModels: the eight (position, reverse) settings in source order, each followed by the asserted name
    names = re.findall(r'std::string\("([FR]{2}[+-])"\), alignmentModelName\(alignmentModel\(f1, f2\)\)', text)
    assert names == ["FF+", "FR+", "RR+", "RF+", "FF-", "FR-", "RR-", "RF-"], names
    settings = [(0, 0, 1, 0), (0, 0, 1, 1), (0, 1, 1, 1), (0, 1, 1, 0), (2, 0, 1, 0), (2, 0, 1, 1), (2, 1, 1, 1), (2, 1, 1, 0)]   # f1.position, f1.reverse, f2.position, f2.reverse as the test sets them
    out["alignment_models"] = [{"f1": [a, b], "f2": [c, d], "name": n} for (a, b, c, d), n in zip(settings, names)]
    out["alignment_classes"] = [{"model": m, "name": n} for n, m in re.findall(r'std::string\("([FR][+-])"\), alignmentClassName\(alignmentClass\(TemplateLengthStatistics::(\w+)\)\)', text)]
    assert len(out["alignment_classes"]) == 8
    # mateOrientation / mateMinPosition / mateMaxPosition: blocks "TemplateLengthStatistics tls(100, 200, 170, 160, 175, M0, M1, -1);" + asserts
    mates = []
    for fn, key in (("mateOrientation", "orientation"), ("mateMinPosition", "min_position"), ("mateMaxPosition", "max_position")):
        body = text[text.index("::test" + fn[0].upper() + fn[1:] + "()"):]
        body = body[:body.index("\nvoid ", 10)] if "\nvoid " in body[10:] else body
        blocks = re.split(r'TemplateLengthStatistics tls\(', body)[1:]
        assert len(blocks) == 8, (fn, len(blocks))
        for blk in blocks:
            args = [a.strip() for a in blk[:blk.index(")")].split(",")]
            stats = [int(a) for a in args[:5]]
            m0, m1 = args[5].split("::")[1], args[6].split("::")[1]
            drift = int(args[7])
            lens = re.search(r'readLengths\[\] = \{(\d+), (\d+)\}', blk)
            for exp, ri, rev, rest in re.findall(r'CPPUNIT_ASSERT_EQUAL\((\w+), tls\.' + fn + r'\((\d), (true|false)([^)]*)\)\)', blk):
                e = {"stats": stats, "models": [m0, m1], "drift": drift, "what": key, "read_index": int(ri), "reverse": rev == "true",
                     "expected": {"true": 1, "false": 0}.get(exp, None) if exp in ("true", "false") else int(exp.rstrip("L"))}
                if lens:
                    e["read_lengths"] = [int(lens.group(1)), int(lens.group(2))]
                    e["position"] = int(rest.split(",")[1])
                mates.append(e)
    assert len(mates) == 3 * 8 * 4, len(mates)
    out["mates"] = mates
    # addTemplates(): the asserted statistics after the first 10000 templates, and the drift variants
    seq = {k: int(v) for k, v in re.findall(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, tls\.getStatistics\(\)\.get(\w+)\(\)\)', text[text.index("::addTemplates"):text.index("::testStatistics")]) and
           [(name, val) for val, name in re.findall(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, tls\.getStatistics\(\)\.get(\w+)\(\)\)', text[text.index("::addTemplates"):text.index("::testStatistics")])]}
    assert seq == {"Min": 14, "Median": 5001, "Max": 9987, "LowStdDev": 3414, "HighStdDev": 3413}, seq
    drift = int(re.search(r'TemplateLengthDistribution tls\((\d+)\);', text[text.index("::testMateDriftRange"):]).group(1))
    out["add_templates"] = {"after_10000": seq, "all_intermediate_results_false": True, "last_result_true": True, "mate_drift_range": drift}
    json.dump(out, open(os.path.join(OUT, "template_length_statistics.json"), "w"), indent=1)
    return len(mates)


def make_clippers():
    """the literal (read, reference) pairs and asserted CIGAR / position of the two end clipper tests"""
    text = strip_comments(open(os.path.join(REF, "testSemialignedClipper.cpp")).read())
    semi = []
    for m in re.finditer(r'fragmentMetadata\.reverse = (true|false);\s*align\("([^"]*)",\s*"([^"]*)",\s*noAdapters,\s*fragmentMetadata\);(.*?)\n}', text, re.S):
        rev, read, ref, asserts = m.groups()
        cigar = re.search(r'std::string\("([0-9A-Z]+)"\), fragmentMetadata\.getCigarString', asserts).group(1)
        pos = int(re.search(r'ReferencePosition\(0, (\d+)U\), fragmentMetadata\.getStrandReferencePosition', asserts).group(1))
        semi.append({"read": read, "reference": ref, "reverse": rev == "true", "cigar": cigar, "position": pos})
    assert len(semi) == 4, len(semi)
    text = strip_comments(open(os.path.join(REF, "testOverlappingEndsClipper.cpp")).read())
    over = []
    for m in re.finditer(r'init\("([^"]*)", "([^"]*)", (true|false),\s*"([^"]*)", "([^"]*)", (true|false),\s*"([^"]*)", templ, contigList\);(.*?)\n    }', text, re.S):
        r1, q1, v1, r2, q2, v2, ref, body = m.groups()
        after = body[body.index("clipper.clip"):]
        cig = re.findall(r'std::string\("([0-9A-Z]+)"\), templ\.getFragmentMetadata\((\d)\)\.getCigarString', after)
        pos = re.findall(r'CPPUNIT_ASSERT_EQUAL\((\d+)L, templ\.getFragmentMetadata\((\d)\)\.position', after)
        over.append({"read1": r1, "quality1": q1, "reverse1": v1 == "true", "read2": r2, "quality2": q2, "reverse2": v2 == "true", "reference": ref,
                     "cigar": [dict((int(i), c) for c, i in cig)[k] for k in (0, 1)], "position": [dict((int(i), int(p)) for p, i in pos)[k] for k in (0, 1)]})
    assert len(over) == 2, len(over)
    json.dump({"source": "lib/alignment/cppunit/testSemialignedClipper.cpp:189-251, testOverlappingEndsClipper.cpp:109-157", "semialigned": semi, "overlapping": over},
              open(os.path.join(OUT, "clippers.json"), "w"), indent=1)
    return len(semi), len(over)


def make_template_builder():
    """testTemplateBuilder.cpp: the candidate lists each test hands to TemplateBuilder::buildTemplate and every value it asserts
    afterwards (template and fragment alignment scores = the MAPQ arithmetic, positions, CIGAR references ...).
    Fixture (BuilderInit.hh:122-132): contigs c0..c4 drawn with glibc rand() in the order c2(230), c4(60), c1(220), c0(210),
    c3 = "AAAAA" + c2.  The stream position at which the fixture is constructed depends on the other fixtures of the test
    binary, so the recipe is evaluated at several positions of the default-seed stream; the asserted values do not depend on it."""
    text = strip_comments(open(os.path.join(REF, "testTemplateBuilder.cpp")).read())
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    fixtures = []
    for _ in range(4):
        draw = lambda n: "".join("ACGT"[libc.rand() % 4] for _ in range(n))
        c2 = draw(230); c4 = draw(60); c1 = draw(220); c0 = draw(210)
        fixtures.append([c0, c1, c2, "AAAAA" + c2, c4])
    # the two literal candidates of the fixture (:121-122)
    def literal(name):
        m = re.search(name + r'\(getFragmentMetadata\(([^)]*)\)\)', text)
        a = [x.strip() for x in m.group(1).split(",")]
        return {"contig_id": int(a[0]), "position": int(a[1]), "observed_length": int(a[2]), "read_index": int(a[3]), "reverse": a[4] == "true",
                "cigar_offset": int(a[5]), "cigar_length": int(a[6]), "mismatch_count": int(a[8]), "log_probability": float(a[9]),
                "unique_seed_count": int(a[10]), "alignment_score": int(a[11])}
    f0_0, f0_1 = literal("f0_0"), literal("f0_1")
    assert (f0_0["position"], f0_0["log_probability"], f0_1["position"], f0_1["log_probability"]) == (2, -8.0, 107, -12.0)
    bcl = re.search(r'bcl0\(getBcl\(readMetadataList, contigList, (\d+), (\d+), (\d+)\)\)', text)
    out = {"source": "lib/alignment/cppunit/testTemplateBuilder.cpp:96-373, BuilderInit.hh:122-169", "fixtures": fixtures,
           "bcl": {"contig": int(bcl.group(1)), "offset0": int(bcl.group(2)), "offset1": int(bcl.group(3))}, "f0_0": f0_0, "f0_1": f0_1, "cases": []}

    def asserts(body):
        """{(fragment index or 't', field): literal} for the CPPUNIT_ASSERT_EQUALs of one stretch of test code"""
        exp = {}
        for lit, who, field in re.findall(r'CPPUNIT_ASSERT_EQUAL\(([^,]+), bamTemplate\.(?:getFragmentMetadata\((\d)\)\.)?(\w+)(?:\(\))?\)', body):
            exp[(who if who else "t", field)] = lit.strip()
        return exp
    def value(lit, best):
        lit = lit.strip()
        m = re.match(r'^(-?\d+)[UL]*$', lit)
        if m: return int(m.group(1))
        if lit in ("true", "false"): return lit == "true"
        m = re.match(r'^(f0_0|f0_1)\.logProbability$', lit)
        if m: return {"f0_0": f0_0, "f0_1": f0_1}[m.group(1)]["log_probability"]
        m = re.match(r'^(?:unsigned|long)?\(?(best[01])\.(\w+)', lit)
        if m:
            b = best[m.group(1)]
            key = {"getFStrandReferencePosition": None, "getObservedLength": "observed_length", "getReadIndex": "read_index", "isReverse": "reverse", "cigarOffset": "cigar_offset",
                   "getCigarLength": "cigar_length", "getMismatchCount": "mismatch_count", "logProbability": "log_probability", "uniqueSeedCount": "unique_seed_count"}[m.group(2)]
            if key is None: return b["contig_id"] if "getContigId" in lit else b["position"]
            return b[key]
        raise ValueError(lit)
    names = {"contigId": "contig_id", "position": "position", "observedLength": "observed_length", "readIndex": "read_index", "reverse": "reverse", "cigarOffset": "cigar_offset",
             "cigarLength": "cigar_length", "mismatchCount": "mismatch_count", "logProbability": "log_probability", "uniqueSeedCount": "unique_seed_count",
             "alignmentScore": "alignment_score", "getAlignmentScore": "alignment_score"}
    def case(name, frags0, frags1, body, best=None):
        exp = {"template_score": None, "fragments": [{}, {}]}
        for (who, field), lit in asserts(body).items():
            if who == "t": exp["template_score"] = value(lit, best)
            else: exp["fragments"][int(who)][names[field]] = value(lit, best)
        out["cases"].append({"name": name, "fragments0": frags0, "fragments1": frags1, "expected": exp})
    fn = lambda n: text[text.index("::" + n + "()"):text.index("\nvoid ", text.index("::" + n + "()") + 10)] if "\nvoid " in text[text.index("::" + n + "()") + 10:] else text[text.index("::" + n + "()"):]
    # testEmptyMatchList: checkUnalignedTemplate + score 0
    unaligned = {"no_match": True, "observed_length": 0, "reverse": False, "cigar_offset": 0, "cigar_length": 0, "mismatch_count": 0, "log_probability": 0.0,
                 "unique_seed_count": 0, "alignment_score": 0xffffffff}
    assert "CPPUNIT_ASSERT_EQUAL(-1U, bamTemplate.getFragmentMetadata(i).alignmentScore)" in text
    out["cases"].append({"name": "empty", "fragments0": [], "fragments1": [], "expected": {"template_score": 0, "fragments": [dict(unaligned, read_index=0), dict(unaligned, read_index=1)]}})
    orphan = fn("testOrphan")
    first, second = orphan.split("// align on the second read only") if "// align on the second read only" in orphan else (None, None)
    if first is None:   # comments were stripped: split at the second buildTemplate call
        idx = [m.start() for m in re.finditer(r'templateBuilder->buildTemplate', orphan)]
        first, second = orphan[idx[0]:idx[1]], orphan[idx[1]:]
    case("orphan_read1", [f0_0], [], first)
    case("orphan_read2", [], [f0_1], second)
    uniq = fn("testUnique")
    case("unique_pair", [f0_0], [f0_1], uniq[uniq.index("templateBuilder->buildTemplate"):])
    # testMultiple: the list construction of :283-336 transcribed
    fr0, fr1 = [], []
    t0, t1 = dict(f0_0), dict(f0_1)
    for _ in range(2):
        fr0.append(dict(t0)); t0["position"] += 56; fr0.append(dict(t0)); t0["position"] += 65; fr1.append(dict(t1)); t1["position"] += 300
    t0, t1 = dict(f0_0, contig_id=1), dict(f0_1, contig_id=1)
    for _ in range(2):
        t0["position"] += 56; fr0.append(dict(t0)); t0["position"] += 65; fr0.append(dict(t0)); t1["position"] += 401; fr1.append(dict(t1))
    t0, t1 = dict(f0_0, contig_id=1), dict(f0_1, contig_id=1)
    t0["log_probability"] += 2; t1["log_probability"] += 2
    fr0.append(dict(t0)); best0 = dict(t0); fr1.append(dict(t1)); best1 = dict(t1)
    t0["log_probability"] -= 2; t1["log_probability"] -= 2
    for _ in range(2):
        t0["position"] += 36; fr0.append(dict(t0)); t0["position"] += 45; fr0.append(dict(t0)); t1["position"] += 402; fr1.append(dict(t1))
    mult = fn("testMultiple")
    assert mult.count("t0.position += 56") == 2 and "t1.position += 401" in mult and "t1.position += 402" in mult and "t0.position += 36" in mult
    case("multiple", fr0, fr1, mult[mult.index("templateBuilder->buildTemplate"):], {"best0": best0, "best1": best1})
    json.dump(out, open(os.path.join(OUT, "template_builder.json"), "w"), indent=1)
    return len(out["cases"]), sum(len(f) for c in out["cases"] for f in c["expected"]["fragments"])


def make_shadow_aligner():
    """testShadowAligner.cpp: four blocks (two tests x two orientations); in each an orphan on read 1 rescues its mate and the mate
    found rescues read 1 back.  Inputs: the getBcl() arguments, the template length statistics and the orphan's strand; expected:
    every asserted field of the two rescued fragments.  Contigs as in the fixture (getContigList(190, 300, 422)), evaluated at four
    positions of the glibc rand() stream."""
    text = strip_comments(open(os.path.join(REF, "testShadowAligner.cpp")).read())
    m = re.search(r'readMetadataList\(getReadMetadataList\((\d+), (\d+)\)\)', text)
    lens = (int(m.group(1)), int(m.group(2)))
    m = re.search(r'contigList\(getContigList\((\d+), (\d+), (\d+)\)\)', text)
    l0, l1, l4 = int(m.group(1)), int(m.group(2)), int(m.group(3))
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    fixtures = []
    for _ in range(4):
        draw = lambda n: "".join("ACGT"[libc.rand() % 4] for _ in range(n))
        c2 = draw(230); c4 = draw(l4); c1 = draw(l1); c0 = draw(l0)
        fixtures.append([c0, c1, c2, "AAAAA" + c2, c4])
    models = {"FFp": 0, "FRp": 1, "RFp": 2, "RRp": 3, "FFm": 4, "FRm": 5, "RFm": 6, "RRm": 7}
    blocks = []
    for blk in re.split(r'const TemplateLengthStatistics tls\(', text)[1:]:
        a = [x.strip() for x in blk[:blk.index(")")].split(",")]
        tls = {"min": int(a[0]), "max": int(a[1]), "median": int(a[2]), "low_std_dev": int(a[3]), "high_std_dev": int(a[4]),
               "model0": models[a[5].split("::")[1]], "model1": models[a[6].split("::")[1]]}
        b = re.search(r'getBcl\(readMetadataList, contigList, (\d+), (\d+), (\d+), (true|false), (true|false)\)', blk)
        orphan_reverse = re.search(r'fragment0\.reverse = (true|false);', blk).group(1) == "true"
        halves = re.split(r'fragment0 = fragment1;', blk)
        exp = []
        for half, who in ((halves[0], "fragment1"), (halves[1], "fragment0")):
            g = lambda pat, conv=int: conv(re.search(pat, half).group(1))
            exp.append({"position": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)L, ' + who + r'\.position\)'),
                        "reverse": re.search(r'CPPUNIT_ASSERT_EQUAL\((true|false), ' + who + r'\.reverse\)', half).group(1) == "true",
                        "observed_length": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, ' + who + r'\.observedLength\)'),
                        "mismatch_count": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, ' + who + r'\.mismatchCount\)'),
                        "cigar_offset": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, ' + who + r'\.cigarOffset\)'),
                        "cigar_length": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, ' + who + r'\.cigarLength\)'),
                        "first_cigar_word": g(r'CPPUNIT_ASSERT_EQUAL\((\d+)U << 4, shadowAligner\.getCigarBuffer\(\)\[0\]\)') << 4,
                        "log_probability": g(r'CPPUNIT_ASSERT_DOUBLES_EQUAL\((-[0-9.]+), ' + who + r'\.logProbability', float),
                        "log_probability_tolerance": g(r'CPPUNIT_ASSERT_DOUBLES_EQUAL\(-[0-9.]+, ' + who + r'\.logProbability, ([0-9.]+)\)', float)})
        blocks.append({"tls": tls, "bcl": {"contig": int(b.group(1)), "offset0": int(b.group(2)), "offset1": int(b.group(3)), "reverse0": b.group(4) == "true", "reverse1": b.group(5) == "true"},
                       "orphan_reverse": orphan_reverse, "expected": exp})
    assert len(blocks) == 4, len(blocks)
    json.dump({"source": "lib/alignment/cppunit/testShadowAligner.cpp:56-252, BuilderInit.hh:122-169", "read_lengths": lens, "fixtures": fixtures, "blocks": blocks},
              open(os.path.join(OUT, "shadow_aligner.json"), "w"), indent=1)
    return len(blocks)


def make_fragment_builder():
    """testFragmentBuilder.cpp: seed matches (SeedId, ReferencePosition) in, FragmentBuilder::build candidates out.  For every test:
    the match list it pushes, the cluster it builds for, and every asserted value (list sizes, CIGAR buffer words, fragment fields,
    log probabilities with the test's tolerance).  Fixture (BuilderInit.hh:122-132, getContigList() defaults): contigs drawn with
    glibc rand() at eight positions of the default-seed stream; the clusters' BCL bytes are evaluated per fixture from the
    constructor's recipes (:41-52).  testMismatches presumes bases of the drawn contig (its own sanity asserts at :383-390, :447);
    `mismatch_fixtures` lists the fixtures for which they hold."""
    text = strip_comments(open(os.path.join(REF, "testFragmentBuilder.cpp")).read())
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[b] for b in reversed(s))
    enc = lambda bases: [(40 << 2) | "ACGT".index(b) for b in bases]
    m = re.search(r'bcl0\(getBcl\(readMetadataList, contigList, (\d+), (\d+), (\d+)\)\)', text); b0 = [int(x) for x in m.groups()]
    m = re.search(r'bcl2\(getBcl\(readMetadataList, contigList, (\d+), (\d+), (\d+)\)\)', text); b2 = [int(x) for x in m.groups()]
    m = re.search(r'bcl3\(subv\(bcl0, 0,(\d+)\) \+\s*"(.)" \+\s*subv\(bcl0, (\d+), (\d+)\) \+\s*"(.)" \+\s*subv\(bcl0, (\d+)\)\)', text)
    assert m and int(m.group(3)) == int(m.group(1)) + 1 and int(m.group(6)) == int(m.group(3)) + int(m.group(4)) + 1
    at0, ch0, at1, ch1 = int(m.group(1)), ord(m.group(2)), int(m.group(3)) + int(m.group(4)), ord(m.group(5))
    assert re.search(r'bcl4l\(getBcl\(substr\(contigList\[4\]\.forward_, 0, 44\) \+ substr\(contigList\[4\]\.forward_, 0, 56\) \+\s*substr\(reverseComplement\(contigList\[4\]\.forward_\), 0, 42\) \+ substr\(reverseComplement\(contigList\[4\]\.forward_\), 0, 58\)\)\)', text)
    assert re.search(r'bcl4t\(getBcl\(substr\(contigList\[4\]\.forward_, 16, 44\) \+ substr\(contigList\[4\]\.forward_, 0, 56\) \+\s*substr\(reverseComplement\(contigList\[4\]\.forward_\), 18, 42\) \+ substr\(reverseComplement\(contigList\[4\]\.forward_\), 0, 58\)\)\)', text)
    assert re.search(r"bcl4lt\(getBcl\(std::vector<char>\(10, 'A'\) \+ contigList\[4\]\.forward_ \+ std::vector<char>\(30, 'C'\) \+\s*std::vector<char>\(15, 'G'\) \+ reverseComplement\(contigList\[4\]\.forward_\) \+ std::vector<char>\(25, 'T'\)\)\)", text)
    tiles = {"tile0": int(re.search(r'tile0\((\d+)\)', text).group(1)), "tile2": int(re.search(r'tile2\((\d+)\)', text).group(1))}
    cluster_ids = {"clusterId0": int(re.search(r'clusterId0\((\d+)\)', text).group(1)), "clusterId2": int(re.search(r'clusterId2\((\d+)\)', text).group(1))}
    fixtures, mismatch_fixtures = [], []
    for k in range(8):
        draw = lambda n: "".join("ACGT"[libc.rand() % 4] for _ in range(n))
        c2 = draw(230); c4 = draw(60); c1 = draw(220); c0 = draw(210)
        contigs = [c0, c1, c2, "AAAAA" + c2, c4]
        def get_bcl(contig, o0, o1):
            f = contigs[contig]
            return enc(f[o0:o0 + 100] + rc(f)[o1:o1 + 100])
        bcl0 = get_bcl(*b0)
        bcl3 = list(bcl0); bcl3[at0] = ch0; bcl3[at1] = ch1
        r4 = rc(c4)
        clusters = {"cluster0": bcl0, "cluster2": get_bcl(*b2), "cluster3": bcl3,
                    "cluster4l": enc(c4[0:44] + c4[0:56] + r4[0:42] + r4[0:58]),
                    "cluster4t": enc(c4[16:60] + c4[0:56] + r4[18:60] + r4[0:58]),
                    "cluster4lt": enc("A" * 10 + c4 + "C" * 30 + "G" * 15 + r4 + "T" * 25)}
        assert all(len(v) == 200 for v in clusters.values())
        fixtures.append({"contigs": contigs, "clusters": {n: bytes(v).hex() for n, v in clusters.items()}})
        # the sanity asserts of testMismatches
        if (bcl0[at0] & 3) != (bcl3[at0] & 3) and (bcl0[at1] & 3) != (bcl3[at1] & 3) and (bcl3[at0] & 3) == 0:
            mismatch_fixtures.append(k)
    assert mismatch_fixtures
    seed_offsets = [0, 32, 64, 0, 32, 64]          # BuilderInit.hh:33-45
    ctor = re.search(r'FragmentBuilder fragmentBuilder\(flowcells, (\d+), seedMetadataList\.size\(\)/2, (\d+), (true|false),', text)
    ops = {"ALIGN": 0, "SOFT_CLIP": 4}
    cases = []
    def body_of(name):
        a = text.index("::" + name + "(")
        b = text.find("\nvoid ", a + 10)
        return text[a:b if b >= 0 else len(text)]
    def one(name, body, env):
        ev = lambda e: int(eval(e, {}, env))
        matches = [{"tile": tiles[t], "cluster": cluster_ids[c], "seed": ev(s), "reverse": r == "true", "contig": int(ct), "position": ev(pos)}
                   for t, c, s, r, ct, pos in re.findall(r'SeedId\((tile\d), 0, (clusterId\d), (\w+), (true|false)\s*\), ReferencePosition\((\d+), ([^)]+)\)', body)]
        cl = re.search(r'matchList\.begin\(\), matchList\.end\(\), (cluster\w+), (true|false)\)', body)
        threshold = int(re.search(r'FragmentBuilder fragmentBuilder\(flowcells, (\d+),', body).group(1))
        exp = {"fragments": {}, "cigar_words": {}, "list_sizes": {}, "cigar_buffer_size": None}
        for n, i in re.findall(r'CPPUNIT_ASSERT_EQUAL\(\(size_t\)(\d+), fragmentBuilder\.getFragments\(\)\[(\d)\]\.size\(\)\)', body): exp["list_sizes"][i] = int(n)
        m = re.search(r'CPPUNIT_ASSERT_EQUAL\(\(size_t\)(\d+), fragmentBuilder\.getCigarBuffer\(\)\.size\(\)\)', body)
        if m: exp["cigar_buffer_size"] = int(m.group(1))
        for ln, op, k in re.findall(r'CPPUNIT_ASSERT_EQUAL\(\(unsigned\)\(\((\d+)<<4\)\|Cigar::(\w+)\), fragmentBuilder\.getCigarBuffer\(\)\[(\d+)\]\)', body):
            w = (int(ln) << 4) | ops[op]
            assert exp["cigar_words"].get(k, w) == w
            exp["cigar_words"][k] = w
        for lit, i, j, field in re.findall(r'CPPUNIT_ASSERT_EQUAL\((?:\(\w+ ?\w*\s?\))?(\w+), fragmentBuilder\.getFragments\(\)\[(\d)\]\[(\d)\]\.(\w+)\)', body):
            v = {"true": 1, "false": 0}.get(lit)
            if v is None: v = int(re.match(r'^(\d+)[UL]*$', lit).group(1))
            exp["fragments"].setdefault(i + "," + j, {})[field] = v
        for lit, i, j, tol in re.findall(r'CPPUNIT_ASSERT_DOUBLES_EQUAL\(\(double\)(-[0-9.]+), fragmentBuilder\.getFragments\(\)\[(\d)\]\[(\d)\]\.logProbability, \(double\)([0-9.]+)\)', body):
            exp["fragments"].setdefault(i + "," + j, {})["logProbability"] = [float(lit), float(tol)]
        cases.append({"name": name, "matches": matches, "cluster": cl.group(1) if cl else None, "with_gaps": (cl.group(2) == "true") if cl else True,
                      "repeat_threshold": threshold, "expected": exp})
    aux = body_of("auxSingleSeed")
    for name in ("testSingleSeed", "testSeedOffset"):
        s0, s1 = [int(x) for x in re.search(r'auxSingleSeed\((\d+), (\d+)\)', body_of(name)).groups()]
        one(name, aux, {"s0": s0, "s1": s1, "offset0": seed_offsets[s0], "offset1": seed_offsets[s1]})
    for name in ("testMultiSeed", "testRepeats", "testMismatches", "testLeadingSoftClips", "testTrailingSoftClips", "testLeadingAndTrailingSoftClips"):
        body = body_of(name)
        env = {}
        for var, val in re.findall(r'const unsigned (s[01]) = (\d+);', body): env[var] = int(val)
        if "s0" in env: env.update(offset0=seed_offsets[env["s0"]], offset1=seed_offsets[env["s1"]])
        one(name, body, env)
    for c in cases:
        assert c["matches"] and c["cluster"] and c["expected"]["fragments"], c["name"]
    n_asserts = sum(len(f) for c in cases for f in c["expected"]["fragments"].values()) + sum(len(c["expected"]["cigar_words"]) + len(c["expected"]["list_sizes"]) for c in cases)
    json.dump({"source": "lib/alignment/cppunit/testFragmentBuilder.cpp:33-598, BuilderInit.hh:33-169", "read_lengths": [100, 100], "seed_offsets": seed_offsets, "seed_length": 32,
               "gapped_mismatches_max": int(ctor.group(2)), "scores": [2, -1, -15, -3, 25], "gap_limit": 20000,
               "fixtures": fixtures, "mismatch_fixtures": mismatch_fixtures, "cases": cases},
              open(os.path.join(OUT, "fragment_builder.json"), "w"), indent=1)
    return len(cases), n_asserts


def make_oligo():
    """The small bit-layout / k-mer tests behind the seed lookup, mate rescue and the index builder:
    testMatchFinderClusterInfo.cpp (ClusterInfo's two bytes), testKmerGenerator.cpp (N-skipping k-mer stream, generateKmer,
    getMaxKmer), testPermutate.cpp (block permutations, the 6 / 70 permutation lists) and testNeighborsFinder.cpp (the 16 k-mer list
    and the flags findNeighbors must set).  Statements are replayed as data: operations in source order with the asserted values."""
    lib = os.path.join(os.path.dirname(REF), "..")
    out = {"source": "lib/alignment/cppunit/testMatchFinderClusterInfo.cpp:38-91, lib/oligo/cppunit/testKmerGenerator.cpp:37-97, "
                     "lib/oligo/cppunit/testPermutate.cpp:41-170, lib/reference/cppunit/testNeighborsFinder.cpp:44-113"}
    # ---- ClusterInfo: a script of constructions, mutations and asserts
    text = strip_comments(open(os.path.join(REF, "testMatchFinderClusterInfo.cpp")).read())
    body = text[text.index("::testFields()"):]
    script = []
    for st in body.split(";"):
        st = " ".join(st.split())
        m = re.search(r'ClusterInfo (\w+)$', st)
        if m and "using" not in st: script.append({"op": "new", "var": m.group(1)}); continue
        m = re.search(r'(\w+)\.markReadComplete\((\d+)\)$', st)
        if m: script.append({"op": "markReadComplete", "var": m.group(1), "arg": int(m.group(2))}); continue
        m = re.search(r'(\w+)\.setBarcodeIndex\((\d+)U\)$', st)
        if m: script.append({"op": "setBarcodeIndex", "var": m.group(1), "arg": int(m.group(2))}); continue
        m = re.search(r'CPPUNIT_ASSERT\((!?)(\w+)\.isBarcodeSet\(\)\)$', st)
        if m: script.append({"op": "assert", "var": m.group(2), "what": "isBarcodeSet", "expected": m.group(1) != "!"}); continue
        m = re.search(r'CPPUNIT_ASSERT_EQUAL\((true|false), (\w+)\.isReadComplete\((\d)\)\)$', st)
        if m: script.append({"op": "assert", "var": m.group(2), "what": "isReadComplete", "arg": int(m.group(3)), "expected": m.group(1) == "true"}); continue
        m = re.search(r'CPPUNIT_ASSERT_EQUAL\((\d+)U, (\w+)\.getBarcodeIndex\(\)\)$', st)
        if m: script.append({"op": "assert", "var": m.group(2), "what": "getBarcodeIndex", "expected": int(m.group(1))}); continue
    assert sum(1 for x in script if x["op"] == "assert") == body.count("CPPUNIT_ASSERT"), (sum(1 for x in script if x["op"] == "assert"), body.count("CPPUNIT_ASSERT"))
    out["cluster_info"] = script
    # ---- KmerGenerator
    text = strip_comments(open(os.path.join(lib, "oligo", "cppunit", "testKmerGenerator.cpp")).read())
    uns = text[text.index("::testUnsigned()"):text.index("::testConstMethods()")]
    streams = []
    for blk in re.findall(r'\{\s*const std::string s(?:\(| = std::string\()"(\w+)"\);(.*?)\n    \}', uns, flags=re.S):
        k = int(re.search(r'kmerGenerator\(v\.begin\(\), v\.end\(\), (\d+)\)', blk[1]).group(1))
        kmers = [int(x, 16) for x in re.findall(r'CPPUNIT_ASSERT_EQUAL\(kmer, (0x[0-9A-Fa-f]+)U\)', blk[1])]
        positions = [int(x) for x in re.findall(r'CPPUNIT_ASSERT_EQUAL\(position - v\.begin\(\), (\d+)L\)', blk[1])]
        n_true = len(re.findall(r'CPPUNIT_ASSERT\(kmerGenerator\.next', blk[1])); n_false = len(re.findall(r'CPPUNIT_ASSERT\(!kmerGenerator\.next', blk[1]))
        assert len(kmers) == len(positions) == n_true and n_false == 1
        streams.append({"sequence": blk[0], "k": k, "kmers": kmers, "positions": positions})
    assert len(streams) == 3
    const = text[text.index("::testConstMethods()"):]
    max_kmers = [[int(k), int(v)] for v, k in re.findall(r'CPPUNIT_ASSERT_EQUAL\((\d+)UL, isaac::oligo::getMaxKmer<unsigned long>\((\d+)\)\)', const)]
    gen = []
    for blk in re.findall(r'\{\s*const std::string s(?:\(| = std::string\()"(\w+)"\);(.*?)\n    \}', const, flags=re.S):
        m = re.search(r'CPPUNIT_ASSERT\((!?)isaac::oligo::generateKmer\((\d+), kmer, s\.begin\(\), s\.end\(\)\)\)', blk[1])
        e = {"sequence": blk[0], "k": int(m.group(2)), "ok": m.group(1) != "!"}
        b = re.search(r'BOOST_BINARY\(([01 ]+)\)', blk[1])
        if b: e["kmer"] = int(b.group(1).replace(" ", ""), 2)
        gen.append(e)
    assert len(max_kmers) == 2 and len(gen) == 2 and "kmer" in gen[1]
    out["kmer_generator"] = {"streams": streams, "max_kmer": max_kmers, "generate_kmer": gen}
    # ---- Permutate
    text = strip_comments(open(os.path.join(lib, "oligo", "cppunit", "testPermutate.cpp")).read())
    blocks = []
    for name in ("testFourBlocks", "testEightBlocks"):
        a = text.index("::" + name + "()"); body = text[a:text.index("\n}", a)]
        consts = {n: int(v, 16) for n, v in re.findall(r'const unsigned long (\w+) = (0x[0-9A-F]+)UL;', body)}
        orders = {n: [int(x) for x in re.findall(r'\((\d+)\)', v)] for n, v in re.findall(r'const std::vector<unsigned> (\w+) = list_of((?:\(\d+\))+);', body)}
        perms = {n: (f, t) for n, f, t in re.findall(r'const oligo::Permutate (\w+)\(blockLength, (\w+), (\w+)\);', body)}
        block_length = int(re.search(r'const unsigned blockLength = (\d+);', body).group(1))
        val = lambda x: consts[x] if x in consts else int(x.rstrip("UL"), 16)
        checks = []
        for pn, ro, arg, exp in re.findall(r'CPPUNIT_ASSERT_EQUAL\((\w+)(\.reorder)?\((\w+)\), (\w+)\);', body):
            checks.append({"from": orders[perms[pn][0]], "to": orders[perms[pn][1]], "reorder": bool(ro), "kmer": "%016x" % val(arg), "expected": "%016x" % val(exp)})
        assert len(checks) == body.count("CPPUNIT_ASSERT_EQUAL"), (name, len(checks))
        blocks.append({"name": name, "block_length": block_length, "checks": checks})
    hexs = lambda x: "%x" % x
    g = lambda pat: re.search(pat, text)
    o16, e16 = int(g(r'ORIGINAL16\((0x[0-9A-F]+)U\)').group(1), 16), int(g(r'EXPECTED16\((0x[0-9A-F]+)U\)').group(1), 16)
    o32, e32 = int(g(r'ORIGINAL\((0x[0-9A-F]+)UL\)').group(1), 16), int(g(r'EXPECTED\((0x[0-9A-F]+)UL\)').group(1), 16)
    m = g(r'ORIGINAL64\(isaac::oligo::LongKmerType\((0x[0-9A-F]+)UL\) << 64 \| isaac::oligo::LongKmerType\((0x[0-9A-F]+)UL\)\)'); o64 = (int(m.group(1), 16) << 64) | int(m.group(2), 16)
    m = g(r'EXPECTED64\(isaac::oligo::LongKmerType\((0x[0-9A-F]+)UL\) << 64 \| isaac::oligo::LongKmerType\((0x[0-9A-F]+)UL\)\)'); e64 = (int(m.group(1), 16) << 64) | int(m.group(2), 16)
    lists = []
    for name in ("testTwoErrors", "testFourErrors"):
        a = text.index("::" + name + "()"); body = text[a:text.index("\n}", a)]
        for kt, errors in re.findall(r'getPermutateList<oligo::(\w+)>\((\d)\)', body):
            size = int(re.search(r'CPPUNIT_ASSERT_EQUAL\((\d+)UL, permutateList\.size\(\)\)', body).group(1))
            bases, o, e = {"ShortKmerType": (16, o16, e16), "KmerType": (32, o32, e32), "LongKmerType": (64, o64, e64)}[kt]
            lists.append({"kmer_bases": bases, "error_count": int(errors), "size": size, "original": hexs(o), "expected": hexs(e)})
    assert len(lists) == 6
    out["permutate"] = {"blocks": blocks, "lists": lists}
    # ---- NeighborsFinder
    text = strip_comments(open(os.path.join(lib, "reference", "cppunit", "testNeighborsFinder.cpp")).read())
    aux = re.findall(r'\((MASK[01])(?:\|(0x[0-9A-F]+)UL)?\)\s*\n', text[text.index("kmerListAux = list_of"):text.index("KmerList kmerList;")])
    flags = {int(i): v == "true" for v, i in re.findall(r'CPPUNIT_ASSERT_EQUAL\((true|false), kmerList\[(\d+)\]\.hasNeighbors\)', text)}
    jobs = int(re.search(r'findNeighbors\(kmerList, (\d+)\)', text).group(1))
    masks = re.findall(r'const isaac::oligo::(\w+) (MASK[01]) = isaac::oligo::\w+\((0x[0-9A-F]+)UL\)( << 64)?;', text)
    assert len(aux) == 16 and len(flags) == 16 and len(masks) == 4
    runs = []
    for kt in ("KmerType", "LongKmerType"):
        mk = {n: int(v, 16) << (64 if sh else 0) for t, n, v, sh in masks if t == kt}
        runs.append({"kmer_bases": 32 if kt == "KmerType" else 64, "kmers": [hexs(mk[n] | (int(lit, 16) if lit else 0)) for n, lit in aux]})
    out["neighbors_finder"] = {"jobs": jobs, "expected": [flags[i] for i in range(16)], "runs": runs}
    json.dump(out, open(os.path.join(OUT, "oligo.json"), "w"), indent=1)
    return len(script), len(streams), sum(len(b["checks"]) for b in blocks), len(lists), len(runs)


def make_sorted_reference():
    """reference/cppunit/testSortedReferenceXml.cpp:30-199: the XML document of the fixture and every value checkContigs / checkMasks
    assert after loading it"""
    text = open(os.path.join(REF, "../../reference/cppunit/testSortedReferenceXml.cpp")).read()
    a = text.index(": xmlString(") + len(": xmlString(")
    literal = text[a:text.index(")\n{", a)]
    xml = "".join(bytes(m, "utf-8").decode("unicode_escape") for m in re.findall(r'"((?:[^"\\]|\\.)*)"', literal))
    field = {"genomicPosition_": "genomic_position", "index_": "index", "name_": "name", "filePath_": "file", "offset_": "offset", "size_": "size", "totalBases_": "total_bases",
             "acgtBases_": "acgt_bases", "karyotypeIndex_": "karyotype_index", "bamSqAs_": "bam_sq_as", "bamSqUr_": "bam_sq_ur", "bamM5_": "bam_m5"}
    contigs = [{}, {}]
    for value, i, name in re.findall(r'CPPUNIT_ASSERT_EQUAL\((.+?), sortedReferenceMetadata\.getContigs\(\)\.at\((\d)\)\.(\w+)\);', text):
        m = re.search(r'"(.*)"', value)
        contigs[int(i)][field[name]] = m.group(1) if m else int(re.match(r"\d+", value).group(0))
    masks = {"count": int(re.search(r"CPPUNIT_ASSERT_EQUAL\((\d+)U, unsigned\(list\.size\(\)\)\)", text).group(1)),
             "last_file": re.search(r'boost::filesystem::path\("([^"]+ABCD-02\.dat)"\),\s*list\.back\(\)\.path', text).group(1),
             "mask_width": int(re.search(r"CPPUNIT_ASSERT_EQUAL\((\d+)U, sortedReferenceMetadata\.getDefaultMaskWidth\(\)\)", text).group(1)), "seed_length": 32}
    out = {"source": "lib/reference/cppunit/testSortedReferenceXml.cpp:30-199 (xmlString and the values checkContigs / checkMasks assert)", "xml": xml, "contigs": contigs, "masks": masks}
    json.dump(out, open(os.path.join(OUT, "sorted_reference.json"), "w"), indent=1)
    return len(contigs), sum(len(c) for c in contigs), masks["count"]


def make_duplicate_filtering():
    """lib/build/cppunit/testDuplicateFiltering.cpp: the index entries of the fixture (constructor initialisers :131-205, dataOffset_ :207-258) and,
    per test, the entries handed to DuplicatePairEndFilter(false) and the ones expected to survive (:268-399).  The fake fragment buffer gives
    the fragment at dataOffset the cluster id dataOffset (fillWithUniqueClusterIdPattern :40-47), tile 0 and barcode 0."""
    text = strip_comments(open(os.path.join(REF, "../../build/cppunit/testDuplicateFiltering.cpp")).read())
    entries = {}
    refpos = lambda contig, position: ((contig + 1) << 41) | (position << 1)           # ReferencePosition(contig, position).value
    mate = r"FragmentIndexMate\(\s*(true|false)\s*,\s*(true|false)\s*,\s*(\d+)\s*,\s*FragmentIndexAnchor\((0x[0-9a-fA-F]+|\d+)\)\)"
    for name, c, pos, anchor, shadow, reverse, storage_bin, mate_anchor, rank in re.findall(
            r"(\w+_)\(ReferencePosition\((\d+),\s*(\d+)\),\s*(?:FragmentIndexAnchor\((0x[0-9a-fA-F]+|\d+)\),\s*)?" + mate + r",\s*(\d+)\)", text):
        entries[name] = {"kind": "rs" if anchor else "f", "f_strand_pos": refpos(int(c), int(pos)), "anchor": int(anchor, 0) if anchor else None,
                         "mate_info": int(shadow == "true") | int(reverse == "true") << 1 | int(storage_bin) << 2, "mate_anchor": int(mate_anchor, 0), "rank": int(rank)}
    for name, n in re.findall(r"(\w+_)\.dataOffset_\s*=\s*(\d+)\s*\*\s*sizeof", text):
        entries[name]["cluster_id"] = int(n)        # dataOffset / sizeof(FragmentHeader) would do as well: only equality and order matter
    assert all("cluster_id" in e for e in entries.values()) and len(entries) == 36
    tests = []
    for test_name, body in re.findall(r"void TestDuplicateFiltering::(test\w+)\(\)\s*\{(.*?)\n\}", text, flags=re.S):
        lists = {v: re.findall(r"\((\w+_)\)", items) for v, items in re.findall(r"std::vector<[^>]+>\s+(\w+)\s*=\s*boost::assign::list_of(.*?);", body, flags=re.S)}
        for inp, exp in re.findall(r"testNoDifferences\((\w+),\s*(\w+)\)", body):
            tests.append({"test": test_name, "input": lists[inp], "expected_unique": lists[exp]})
    out = {"source": "lib/build/cppunit/testDuplicateFiltering.cpp:131-399", "entries": entries, "cases": tests}
    json.dump(out, open(os.path.join(OUT, "duplicate_filtering.json"), "w"), indent=1)
    return len(entries), len(tests)


def _realign_fixture(read, ref, gaps, low_clipped, high_clipped):
    """initFragment (testGapRealigner.cpp:175-289) and addGaps (:291-351) of the test fixture, re-run on the test's strings: the concrete
    fragment (forward-strand read, position, CIGAR, observed length, edit distance), reference and gap list GapRealigner::realign is given"""
    ALIGN, INSERT, DELETE, SOFT_CLIP = 0, 1, 2, 4
    left_clipped, right_clipped = low_clipped, high_clipped            # the test's fragments are forward-strand
    unclipped_pos = next(i for i, ch in enumerate(read) if ch != " ")
    left_overhang = next(i for i, ch in enumerate(ref) if ch != " ")
    f_strand_pos = left_clipped + unclipped_pos
    assert f_strand_pos < len(ref)
    cigar = []
    ri = fi = unclipped_pos                                             # readIterator, refIterator as indexes
    edit_distance = 0
    if left_clipped + left_overhang:
        cigar.append((left_clipped + left_overhang, SOFT_CLIP))
        while ri != f_strand_pos + left_overhang and fi != len(ref):
            ri += 1; fi += 1
    observed = 0
    bit = [0, ALIGN]
    read_end = len(read) - right_clipped
    while ri != read_end and fi != len(ref):
        if read[ri] == "-":
            if bit[1] == DELETE:
                bit[0] += 1
            else:
                cigar.append(tuple(bit)); bit = [1, DELETE]
            assert ref[fi] != "*"
            edit_distance += 1; observed += 1
        elif ref[fi] == "*":
            if bit[1] == INSERT:
                bit[0] += 1
            else:
                if bit[0]:
                    cigar.append(tuple(bit))
                bit = [1, INSERT]
            edit_distance += 1
        else:
            if bit[1] == ALIGN:
                bit[0] += 1
            else:
                cigar.append(tuple(bit)); bit = [1, ALIGN]
            edit_distance += ref[fi] != read[ri]
            observed += 1
        ri += 1; fi += 1
    if bit[0]:
        cigar.append(tuple(bit))
    if fi == len(ref):
        if ri != len(read):
            cigar.append((len(read) - ri, SOFT_CLIP))
    elif right_clipped:
        cigar.append((right_clipped, SOFT_CLIP))
    bases = [ch for ch in read if ch in "ACGTN"]                        # TestFragmentAccessor (:116-151)
    contig = "".join(ch for ch in ref if ch not in "* ")
    # addGaps: positions count reference characters that are not '*'
    found = []
    pos, length = 0, 0
    fi = 0
    for gi, g in enumerate(gaps):
        if fi == len(ref):
            if length:
                found.append((pos - abs(length), length))
            pos, length, fi = 0, 0, 0
        if ref[fi] == "*":
            assert g == " "
            fi += 1
            continue
        if g != "*":
            if length < 0:
                found.append((pos + length, length)); length = 0
        else:
            length -= 1
        if g != "-":
            if length > 0:
                found.append((pos - length, length)); length = 0
        else:
            length += 1
        pos += 1
        fi += 1
    if length:
        found.append((pos - abs(length), length))
    return {"read_bases": "".join(bases), "contig": contig, "f_strand_position": f_strand_pos, "cigar": [(n << 4) | op for n, op in cigar], "observed_length": observed,
            "edit_distance": int(edit_distance), "low_clipped": low_clipped, "high_clipped": high_clipped, "gaps": found}


def make_gap_realigner():
    """lib/build/cppunit/testGapRealigner.cpp:456-1733: every realign(...) call of testFull / testMore with the values asserted on its result"""
    text = strip_comments(open(os.path.join(REF, "../../build/cppunit/testGapRealigner.cpp")).read())
    cases = []
    for m in re.finditer(r"const RealignResult result = realign\(", text):
        args_text, end = find_call(text, m.end())
        args = split_args(args_text)
        block_start, block_end = enclosing_block(text, m.start())
        block = text[block_start:block_end]
        costs = (1, 0)
        if args[0][0] != "str":
            costs = (int(args[0][1]), int(args[1][1])); args = args[2:]
        assert all(a[0] == "str" for a in args[:3])
        read, ref, gaps = args[0][1], args[1][1], args[2][1]
        args = [a[1] for a in args]
        low = high = 0
        bin_start, bin_end = 0, None
        if len(args) > 3:
            init = args[3].strip()
            if init != "io::FragmentHeader()":
                before = text[block_start:m.start()]
                for field, value in re.findall(re.escape(init) + r"\.(lowClipped_|highClipped_)\s*=\s*(\d+)", before):
                    if field == "lowClipped_":
                        low = int(value)
                    else:
                        high = int(value)
            if len(args) > 4:
                bin_start = int(re.search(r"ReferencePosition\(0,\s*(\d+)\)", args[4]).group(1))
            if len(args) > 5:
                bin_end = int(re.search(r"ReferencePosition\(0,\s*(\d+)\)", args[5]).group(1))
        after = text[m.end():block_end]
        nxt = after.find("const RealignResult result = realign(")
        if nxt >= 0:
            after = after[:nxt]
        expected = {}
        for value, field in re.findall(r"CPPUNIT_ASSERT_EQUAL\((.+?),\s*(?:int\()?result\.(\w+(?:\.\w+\(\d*\))?)\)?\);", after):
            value = value.strip()
            if "std::string" in value:
                expected[field] = re.search(r'"(.*)"', value).group(1)
            elif "ReferencePosition" in value:
                expected[field] = int(re.search(r"ReferencePosition\(0,\s*(\d+)\)", value).group(1))
            else:
                expected[field] = int(value.rstrip("U"), 0)
        case = _realign_fixture(read, ref, gaps, low, high)
        case.update({"mismatch_cost": costs[0], "gap_open_cost": costs[1], "bin_start": bin_start, "bin_end": bin_end, "expected": expected})
        cases.append(case)
    out = {"source": "lib/build/cppunit/testGapRealigner.cpp:456-1733 (fixture recipe :53-424 re-run on the test's strings)",
           "realigner": {"vigorous": True, "dodgy": False, "gaps_per_fragment": 8, "gap_extend_cost": 0, "clip_semialigned": False}, "cases": cases}
    json.dump(out, open(os.path.join(OUT, "gap_realigner.json"), "w"), indent=1)
    return len(cases), sum(len(c["expected"]) for c in cases)


if __name__ == "__main__":
    print("gap_realigner cases, asserted values:", make_gap_realigner())
    print("duplicate_filtering entries, filter runs:", make_duplicate_filtering())
    print("sorted_reference contigs, asserted contig fields, masks:", make_sorted_reference())
    print("oligo (cluster info steps, k-mer streams, permutate checks, permutation lists, neighbour runs):", make_oligo())
    print("fragment_builder cases, asserted values:", make_fragment_builder())
    print("shadow_aligner blocks:", make_shadow_aligner())
    print("template_builder cases, asserted fragment fields:", make_template_builder())
    print("clippers:", make_clippers())
    print("template_length_statistics asserts:", make_template_length_statistics())
    print("simple_indel cases:", make_simple_indel())
    print("fragment_builder2 cases:", make_fragment_builder2())
    print("sequencing_adapter cases:", make_sequencing_adapter())
    print("bsw cases:", make_bsw())
    print("seed_id:", make_seed_id())
