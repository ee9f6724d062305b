"""The oracle against the reference's own golden vectors (CPU only).

Every vector in tests/golden/*.json is transcribed from a unit test of the reference; the source
file:line is recorded in the JSON.  These tests are what pins the oracle (DESIGN.md "Oracle").
"""
import json
import os

import numpy as np
import pytest

from dentist_amd import sim
from oracle import pyoracle as oz

GOLD = os.path.join(os.path.dirname(__file__), "golden")
OPS = {"sub": 0, "del": 1, "ins": 2}


def render(ref, qry, ops, width):
    """SequenceAlignment.toString (source/dentist/util/string.d:358-418)."""
    rl, cl, ql = [], [], []
    i = j = 0
    for op in ops:
        if op == 0:
            rl.append(ref[i]); cl.append("|" if ref[i] == qry[j] else "*"); ql.append(qry[j]); i += 1; j += 1
        elif op == 1:
            rl.append(ref[i]); cl.append(" "); ql.append("-"); i += 1
        else:
            rl.append("-"); cl.append(" "); ql.append(qry[j]); j += 1
    rl, cl, ql = "".join(rl), "".join(cl), "".join(ql)
    if width == 0:
        return "\n".join((rl, cl, ql))
    chunks = [(rl[k:k + width], cl[k:k + width], ql[k:k + width]) for k in range(0, len(rl), width)]
    return "\n\n".join("\n".join(c) for c in chunks)


def _nw_cases():
    with open(os.path.join(GOLD, "nw_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _nw_cases(), ids=lambda c: f"string.d:{c['line']}")
def test_nw_golden(case):
    ref = np.frombuffer(case["ref"].encode(), dtype=np.uint8)
    qry = np.frombuffer(case["qry"].encode(), dtype=np.uint8)
    score, ops = oz.nw(ref, qry, case["indel"], case["free_shift"])
    if "score" in case:
        assert score == case["score"]
    if "ops" in case:
        assert [int(x) for x in ops] == [OPS[o] for o in case["ops"]]
    assert render(case["ref"], case["qry"], ops, case["width"]) == case["text"]
    # the edit path consumes both sequences exactly
    assert sum(1 for o in ops if o != 2) == len(ref) and sum(1 for o in ops if o != 1) == len(qry)


def parse_ladump(lines):
    """LAdump text (as fed to `dumpLA`, dazzler.d:6431-6443) -> records + trace."""
    chain = {">": 0x4 | 0x10, "-": 0x8, "+": 0x4, ".": 0}
    recs, trace, cur = [], [], None
    it = iter(lines)
    tspace = None
    for ln in it:
        p = ln.split()
        if p[0] == "X":
            tspace = int(p[1])
        elif p[0] == "P":
            cur = {"aread": int(p[1]) - 1, "bread": int(p[2]) - 1,
                   "flags": (1 if p[3] == "c" else 0) | chain[p[4]]}
        elif p[0] == "C":
            cur.update(abpos=int(p[1]), aepos=int(p[2]), bbpos=int(p[3]), bepos=int(p[4]))
        elif p[0] == "T":
            n = int(p[1])
            tps = [tuple(int(x) for x in next(it).split()) for _ in range(n)]
            cur.update(tlen=2 * n, diffs=sum(t[0] for t in tps), toff=len(trace))
            for d, b in tps:
                trace += [d, b]
            recs.append(cur)
    las = np.zeros(len(recs), dtype=oz.LA_DTYPE)
    for i, r in enumerate(recs):
        for k, v in r.items():
            las[i][k] = v
    return las, np.asarray(trace, dtype=np.uint16), tspace


def dentist_flags(f):
    """fillInOverlapHead, source/dentist/dazzler.d:1728-1758."""
    out = []
    if f & 0x20: out.append("disabled")
    if f & 0x1: out.append("complement")
    if (f & 0x4) and not (f & 0x10): out.append("alternateChain")
    if f & 0x8: out.append("chainContinuation")
    if not (f & (0x4 | 0x10 | 0x8)): out.append("unchained")
    return out


@pytest.mark.parametrize("tspace", [100, 126])
def test_las_codec_golden(tmp_path, tspace):
    with open(os.path.join(GOLD, "las_dump.json")) as f:
        g = json.load(f)
    las, trace, ts = parse_ladump(g["dump"])
    assert ts == g["tspace"]
    path = str(tmp_path / "t.las")
    oz.las_write(path, las, trace, tspace)
    raw = open(path, "rb").read()
    # header 12 bytes, 40 bytes per record, 1 or 2 bytes per trace value (dazzler.d:1665-1689, 2019-2025)
    assert len(raw) == 12 + 40 * len(las) + len(trace) * (1 if tspace <= 125 else 2)
    assert int.from_bytes(raw[:8], "little") == len(las) and int.from_bytes(raw[8:12], "little") == tspace
    las2, trace2, ts2 = oz.las_read(path)
    assert ts2 == tspace
    assert len(las2) == len(g["expected"])
    for la, exp in zip(las2, g["expected"]):
        assert [la["aread"] + 1, la["abpos"], la["aepos"]] == exp["a"]
        assert [la["bread"] + 1, la["bbpos"], la["bepos"]] == exp["b"]
        assert sorted(dentist_flags(int(la["flags"]))) == sorted(exp["flags"])
        tp = trace2[la["toff"]:la["toff"] + la["tlen"]].reshape(-1, 2).tolist()
        assert tp == exp["tp"]
        assert la["diffs"] == sum(t[0] for t in exp["tp"])


def test_las_truncated_is_an_error(tmp_path):
    with open(os.path.join(GOLD, "las_dump.json")) as f:
        g = json.load(f)
    las, trace, _ = parse_ladump(g["dump"])
    path = str(tmp_path / "t.las")
    oz.las_write(path, las, trace, 100)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-3])
    with pytest.raises(IOError):
        oz.las_read(path)


def test_trace_translation_golden():
    import ctypes
    with open(os.path.join(GOLD, "trace_cases.json")) as f:
        g = json.load(f)
    la = g["la"]
    tr = np.asarray(la["tp"], dtype=np.uint16).reshape(-1)
    ntp = len(la["tp"])
    L = oz.lib()

    def translate(apos, mode):
        a, b = ctypes.c_int32(), ctypes.c_int32()
        L.oz_translate_trace_point_a(la["abpos"], la["aepos"], la["bbpos"], la["tspace"], tr.ctypes.data,
                                     ntp, apos, {"floor": 0, "ceil": 1}[mode], ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    for apos, ea, eb in g["floor_cases"]:
        assert translate(apos, "floor") == (ea, eb)
    for p1, m1, p2, m2 in g["equal_pairs"]:
        assert translate(p1, m1) == translate(p2, m2)
    # cropToTracePoint (base.d:993-1129) in terms of the translation: front keeps [begin, tp], back [tp, end]
    def crop(seed, pos, mode):
        a, b = translate(pos, mode)
        empty = (a == la["abpos"] or b == la["bbpos"]) if seed == "front" else (a == la["aepos"] or b == la["bepos"])
        return empty, a, b
    for seed, pos, mode, disabled, ea, eb in g["crop_cases"]:
        empty, a, b = crop(seed, pos, mode)
        assert empty == disabled and (disabled or (a, b) == (ea, eb))
    for seed, p1, m1, p2, m2 in g["equal_crop_pairs"]:
        assert crop(seed, p1, m1) == crop(seed, p2, m2)
    s = g["second"]
    n = L.oz_trace_points_up_to_a(s["abpos"], s["aepos"], s["tspace"], len(s["tp"]), s["apos"], 1)
    assert 0 <= n <= len(s["tp"])
    # contigB variant: positions are the running sums of bbases
    tb = np.asarray(la["tp"], dtype=np.uint16).reshape(-1)
    assert L.oz_trace_points_up_to_b(0, 2158, tb.ctypes.data, ntp, 0, 0) == 0
    assert L.oz_trace_points_up_to_b(0, 2158, tb.ctypes.data, ntp, 2158, 0) == ntp
    assert L.oz_trace_points_up_to_b(0, 2158, tb.ctypes.data, ntp, 23, 1) == 1
    assert L.oz_trace_points_up_to_b(0, 2158, tb.ctypes.data, ntp, 24, 0) == 1
    assert L.oz_trace_points_up_to_b(0, 2158, tb.ctypes.data, ntp, 24, 1) == 2


def test_fixture_inputs_match_reference_md5():
    """tests/test-commands.sh:54-61 pins md5(data/assembly-reference.fasta)."""
    import hashlib
    raw = open(os.path.join(GOLD, "test_commands_assembly_reference.fasta"), "rb").read()
    assert hashlib.md5(raw).hexdigest() == "7d6102250532133377d5eeb94bccec59"
    gap = open(os.path.join(GOLD, "test_commands_gap_seq.txt")).read().strip()
    seq = "".join(raw.decode().split("\n")[1:])
    assert len(seq) == 4097 and seq.find(gap) == 2000 and len(gap) == 97


def test_oracle_mapping_against_truth():
    """The restated aligner recovers the simulated placements and honours the trace invariants."""
    from helpers import check_trace_invariants
    w = sim.Workload(200_000, 3, 300, 4000, seed=11, spacing=15000)
    o = oz.default_opts(width=62)
    las, trace, st = oz.align_db(w.contigs, w.reads, o, nthreads=4)
    assert len(las) >= w.reads.n
    check_trace_invariants(las, trace, 100)
    mapped = set()
    for la in las:
        s, e, strand = w.read_truth[la["bread"]]
        cs = w.contig_start[la["aread"]]
        assert (la["flags"] & 1) == strand
        assert cs + la["abpos"] >= s - 60 and cs + la["aepos"] <= e + 60
        mapped.add(int(la["bread"]))
    assert len(mapped) == w.reads.n
    err = las["diffs"].sum() / (las["aepos"] - las["abpos"]).sum()
    assert 0.11 < err < 0.15


def test_consensus_known_answer_of_the_reference():
    """dazzler.d:4257-4299 (unittest of getConsensus): three 1050 bp reads, two of them with one
    substitution each; the consensus of the pile of read 1 must equal the clean third read."""
    from dentist_amd import sim
    d = json.load(open(os.path.join(GOLD, "consensus_3reads.json")))
    db = sim.SeqDb.from_list([sim.encode(r["sequence"].lower()) for r in d["reads"]])
    o = oz.default_opts(skip_self=2, tspace=100, min_len=d["daligner_min_alignment_length"], max_la=64, max_cand=128,
                        width=30)
    las, trace, _ = oz.align_db(db, db, o, nthreads=2)
    assert len(las) == 6 and all(l["aepos"] - l["abpos"] == 1050 for l in las)
    for ref in range(3):
        cons = oz.consensus(db.seq(ref), db, las, trace, ref, 100)
        assert sim.decode(cons) == d["expected_consensus"].lower()


def test_cropping_slice_golden():
    """cropper.d:552-646: the read interval kept by getCroppingSlice for a crop point on the contig
    (complement + seed back: [0, |read| - b), forward + seed front: [0, b) with b = the read
    coordinate of the trace point)."""
    from oracle import process as pr
    g = json.load(open(os.path.join(GOLD, "crop_cases.json")))
    for c in g["cases"]:
        tr = np.asarray(c["tp"], dtype=np.uint16).reshape(-1)
        la = {"abpos": c["abpos"], "aepos": c["aepos"], "bbpos": c["bbpos"]}
        assert int(tr[1::2].sum()) == c["bepos"] - c["bbpos"] and int(tr[0::2].sum()) == c["diffs"]
        for apos, (b0, b1) in c["crops"]:
            _, b = pr.translate_floor(la, tr, apos, 100)
            got = (0, c["read_len"] - b) if c["complement"] else (0, b)
            assert got == (b0, b1), (apos, got)
