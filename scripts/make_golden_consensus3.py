#!/usr/bin/env python3
"""Transcribes the known-answer vector of the reference's getConsensus unit test
(source/dentist/dazzler.d:4257-4299: three 1050 bp reads, two of them with one / two edits; the
consensus of the pile must equal the clean third read) into tests/golden/consensus_3reads.json.
Run in the build container only (reads /root/reference)."""
import json
import os
import re

src = open("/root/reference/source/dentist/dazzler.d").read().split("\n")
block = "\n".join(src[4256:4270])
recs = re.findall(r'">(Sim/\d/0_1050 RQ=0\.975)\\n([a-zA-Z\\n]+)"', block)
assert len(recs) == 3, len(recs)
out = {"source": "source/dentist/dazzler.d:4257-4299 (unittest of getConsensus)",
       "daligner_min_alignment_length": 15,
       "reads": [{"header": h, "sequence": s.replace("\\n", "")} for h, s in recs]}
out["expected_consensus"] = out["reads"][2]["sequence"]
assert all(len(r["sequence"]) == 1050 for r in out["reads"])
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "consensus_3reads.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
