#!/usr/bin/env python3
"""Transcribes the DATA of the reference's pile-up fixture (source/dentist/common/binio/_testdata/
pileupdb.d:27-31, 33-...: 2 pile-ups / 5 read alignments / 7 seeded alignments / 8 local alignments /
393 trace points, realistic damapper chains) into tests/golden/pileupdb_chains.json.  Only numbers,
flags and seeds are kept (vectors, not source text).  Run in the build container:
    python scripts/make_golden_pileupdb.py"""
import json
import os
import re

SRC = "/root/reference/source/dentist/common/binio/_testdata/pileupdb.d"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pileupdb_chains.json")

text = open(SRC).read()
body = text[text.index("return [", text.index("PileUp[] getPileUpsTestData()")):]
tok = re.findall(r"ReadAlignment\(|SeededAlignment\(|AlignmentChain\(|Contig\(\s*\d+\s*,\s*\d+\s*\)|AlignmentFlags\([a-z, ]*\)|"
                 r"LocalAlignment\(|Locus\(\s*\d+\s*,\s*\d+\s*\)|TracePoint\(\s*\d+\s*,\s*\d+\s*\)|AlignmentLocationSeed\.\w+|"
                 r"\[|\]|\d+", body)
piles, depth, i = [], 0, 0
cur_pile = cur_ra = cur_sa = cur_la = None
nums = lambda s: [int(x) for x in re.findall(r"\d+", s)]  # noqa: E731
while i < len(tok):
    t = tok[i]
    if t == "[":
        depth += 1
        if depth == 2:
            cur_pile = []
            piles.append(cur_pile)
    elif t == "]":
        depth -= 1
        if depth == 0:
            break
    elif t == "ReadAlignment(":
        cur_ra = []
        cur_pile.append(cur_ra)
    elif t == "SeededAlignment(":
        cur_sa = {"las": []}
        cur_ra.append(cur_sa)
    elif t == "AlignmentChain(":
        cur_sa["id"] = int(tok[i + 1])
        i += 1
    elif t.startswith("Contig("):
        key = "contigA" if "contigA" not in cur_sa else "contigB"
        cur_sa[key] = nums(t)
    elif t.startswith("AlignmentFlags("):
        cur_sa["complement"] = "complement" in t
    elif t == "LocalAlignment(":
        a, b = nums(tok[i + 1]), nums(tok[i + 2])
        cur_la = {"a": a, "b": b, "diffs": int(tok[i + 3]), "tp": []}
        cur_sa["las"].append(cur_la)
        i += 3
    elif t.startswith("TracePoint("):
        cur_la["tp"].append(nums(t))
    elif t.startswith("AlignmentLocationSeed."):
        cur_sa["seed"] = t.split(".")[1]
    i += 1
counts = dict(pileUps=len(piles), readAlignments=sum(len(p) for p in piles),
              seededAlignments=sum(len(r) for p in piles for r in p),
              localAlignments=sum(len(s["las"]) for p in piles for r in p for s in r),
              tracePoints=sum(len(l["tp"]) for p in piles for r in p for s in r for l in s["las"]))
assert counts == dict(pileUps=2, readAlignments=5, seededAlignments=7, localAlignments=8, tracePoints=393), counts
json.dump({"_source": "source/dentist/common/binio/_testdata/pileupdb.d (getPileUpsTestData; counts :27-31); "
                      "expected file size = sum of T.sizeof x count, binio/pileupdb.d:439-446",
           "counts": counts, "trace_point_distance": 100, "pile_ups": piles}, open(OUT, "w"), indent=0)
print(OUT, counts)
