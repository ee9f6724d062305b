"""pile-ups.db / insertions.db codecs against the reference's own fixture data
(source/dentist/common/binio/_testdata/pileupdb.d, transcribed by scripts/make_golden_pileupdb.py):
file size and index pointers as the reference's unit tests compute them from the D struct sizes
(binio/pileupdb.d:439-446, 505-526), round trip, corruption is an error.  The 393 realistic trace
points are also pushed through the product's trace translation.  CPU only (host code)."""
import json
import os
import struct

import numpy as np
import pytest

import dentist_amd

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SIZEOF = dict(index=48, pile=16, ra=16, seeded=56, la=40, tp=4)   # x86-64 D struct sizes (pileupdb.d:845-897)


def load_fixture():
    g = json.load(open(os.path.join(GOLD, "pileupdb_chains.json")))
    pc, rc, sa, la, tp = [], [], [], [], []
    for pile in g["pile_ups"]:
        pc.append(len(pile))
        for ra in pile:
            rc.append(len(ra))
            for s in ra:
                sa.append((s["id"], s["contigA"][0], s["contigA"][1], s["contigB"][0], s["contigB"][1],
                           1 if s["complement"] else 0, {"front": 0, "back": 1}[s["seed"]], 0, len(s["las"])))
                for l in s["las"]:
                    la.append((l["a"][0], l["a"][1], l["b"][0], l["b"][1], l["diffs"], len(l["tp"])))
                    for d, b in l["tp"]:
                        tp += [d, b]
    return (g, np.asarray(pc, np.int32), np.asarray(rc, np.int32), np.asarray(sa, dtype=dentist_amd.SEEDED_DTYPE),
            np.asarray(la, dtype=dentist_amd.CHAIN_LA_DTYPE), np.asarray(tp, np.uint16))


def test_pileupdb_size_pointers_and_round_trip(tmp_path):
    g, pc, rc, sa, la, tp = load_fixture()
    c = g["counts"]
    path = str(tmp_path / "pile-ups.db")
    dentist_amd.pileupdb_write(path, pc, rc, sa, la, tp)
    raw = open(path, "rb").read()
    total = (SIZEOF["index"] + SIZEOF["pile"] * c["pileUps"] + SIZEOF["ra"] * c["readAlignments"] +
             SIZEOF["seeded"] * c["seededAlignments"] + SIZEOF["la"] * c["localAlignments"] + SIZEOF["tp"] * c["tracePoints"])
    assert len(raw) == total == 2444
    ix = struct.unpack("<6Q", raw[:48])
    assert ix[0] == 48 and ix[1] == ix[0] + 16 * 2 and ix[2] == ix[1] + 16 * 5 and ix[3] == ix[2] + 56 * 7
    assert ix[4] == ix[3] + 40 * 8 and ix[5] == ix[4] + 4 * 393 == len(raw)
    # first pile-up = ArrayStorage(ptr to its read alignments, 2); first seeded alignment of the fixture
    assert struct.unpack("<2Q", raw[48:64]) == (ix[1], 2)
    sid, aid, alen, bid, blen, flags = struct.unpack("<Q4IB", raw[ix[2]:ix[2] + 25])
    assert (sid, aid, alen, bid, blen, flags) == (9, 1, 8300, 1539, 6414, 1)
    lptr, llen, tpd, seed = struct.unpack("<2QHB", raw[ix[2] + 32:ix[2] + 51])
    assert (lptr, llen, tpd, seed) == (ix[3], 1, 0, 0)
    back = dentist_amd.pileupdb_read(path)
    assert np.array_equal(back["pile_counts"], pc) and np.array_equal(back["ra_counts"], rc)
    assert np.array_equal(back["seeded"], sa) and np.array_equal(back["las"], la) and np.array_equal(back["trace"], tp)
    # truncated / inconsistent files are errors
    open(path, "wb").write(raw[:-3])
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.pileupdb_read(path)
    bad = bytearray(raw)
    bad[48:56] = struct.pack("<Q", ix[1] + 16)
    open(path, "wb").write(bytes(bad))
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.pileupdb_read(path)


def test_realistic_chains_through_the_trace_arithmetic():
    """Every local alignment of the fixture: the invariants of base.d:434-458 and the product's
    translateTracePoint at every trace point (floor and ceil) against the running sums."""
    g, pc, rc, sa, la, tp = load_fixture()
    ts, at = g["trace_point_distance"], 0
    for l in la:
        tr = tp[at:at + 2 * l["ntp"]]
        at += 2 * l["ntp"]
        assert int(tr[0::2].sum()) == l["diffs"] and int(tr[1::2].sum()) == l["b_end"] - l["b_begin"]
        assert l["ntp"] == -(-int(l["a_end"]) // ts) - int(l["a_begin"]) // ts
        rec = np.zeros(1, dtype=dentist_amd.LA_DTYPE)[0]
        rec["abpos"], rec["aepos"], rec["bbpos"], rec["bepos"] = l["a_begin"], l["a_end"], l["b_begin"], l["b_end"]
        rec["diffs"], rec["tlen"], rec["toff"] = l["diffs"], 2 * l["ntp"], 0
        b = int(l["b_begin"])
        bounds = [int(l["a_begin"])] + [x for x in range((int(l["a_begin"]) // ts + 1) * ts, int(l["a_end"]), ts)] + [int(l["a_end"])]
        for i, a in enumerate(bounds):
            if i > 0:
                b += int(tr[2 * (i - 1) + 1])
            second = (int(l["a_begin"]) // ts + 1) * ts
            # tracePointsUpTo (base.d:207-244) as written: floor returns 0 for every position below the
            # second trace point -- also for the END of an alignment that lies inside one tile
            exp_floor = (int(l["a_begin"]), int(l["b_begin"])) if a < second else (a, b)
            assert dentist_amd.translate_trace_point(rec, tr, ts, a, "floor") == exp_floor
            assert dentist_amd.translate_trace_point(rec, tr, ts, a, "ceil") == (a, b)
            if a + 1 < int(l["a_end"]) and (a + 1) % ts:
                assert dentist_amd.translate_trace_point(rec, tr, ts, a + 1, "floor") == (a, b)
    assert at == len(tp)


def test_insertiondb_round_trip_and_base_packing(tmp_path):
    g, pc, rc, sa, la, tp = load_fixture()
    # two insertions: a gap closed between contigs 1 and 2 (two overlaps) and an extension (one overlap)
    ins = np.zeros(2, dtype=dentist_amd.INSERTION_REC_DTYPE)
    ins[0]["start_contig"], ins[0]["start_part"], ins[0]["end_contig"], ins[0]["end_part"] = 1, 2, 2, 1
    ins[0]["seq_len"], ins[0]["contig_len"], ins[0]["noverlaps"], ins[0]["nread_ids"] = 41, 0, 2, 3
    ins[1]["start_contig"], ins[1]["start_part"], ins[1]["end_contig"], ins[1]["end_part"] = 2, 2, 2, 3
    ins[1]["seq_len"], ins[1]["noverlaps"], ins[1]["nread_ids"] = 7, 1, 1
    seq = "atgccaactactttgaacgcgccgcaaggcacaggtgcgcct" [:41] + "gattaca"      # testSequence of binio/common.d:407-414
    bases = np.asarray(["acgt".index(c) for c in seq], dtype=np.uint8)
    overlaps = sa[2:5].copy()
    nla = int(overlaps["nla"].sum())
    la0 = int(sa[:2]["nla"].sum())
    las = la[la0:la0 + nla]
    t0 = 2 * int(la[:la0]["ntp"].sum())
    trace = tp[t0:t0 + 2 * int(las["ntp"].sum())]
    path = str(tmp_path / "insertions.db")
    dentist_amd.insertiondb_write(path, ins, bases, [5, 9, 1539, 77], overlaps, las, trace)
    raw = open(path, "rb").read()
    ix = struct.unpack("<7Q", raw[:56])
    assert ix[0] == 56 and ix[1] == 56 + 104 * 2 and ix[2] == ix[1] + 11 + 2 and ix[3] == ix[2] + 56 * 3
    assert ix[6] == len(raw) == ix[5] + 4 * 4
    # a=0 c=1 t=2 g=3, first base in the low bits (binio/common.d:324-345): "atgc" -> 0b01_11_10_00
    assert raw[ix[1]] == 0b01111000
    back = dentist_amd.insertiondb_read(path)
    assert np.array_equal(back["insertions"], ins) and np.array_equal(back["bases"], bases)
    assert back["read_ids"].tolist() == [5, 9, 1539, 77]
    assert np.array_equal(back["seeded"], overlaps) and np.array_equal(back["las"], las) and np.array_equal(back["trace"], trace)
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.insertiondb_write(path, ins, np.full(48, 4, np.uint8), [5, 9, 1539, 77], overlaps, las, trace)


def test_merge_insertions_is_a_keyed_merge_of_the_batch_files(tmp_path):
    """`dentist merge-insertions` (commands/mergeInsertions.d:42-164): batch files merged by (start, end) node; a batch
    that is not sorted is sorted first (:66-72); equal keys keep the order of the files (:120-138, the unittest's
    [1,3,5] + [2,3,4] + [3,5,6] -> [1,2,3,3,3,4,5,5,6] with contig ids as keys)."""
    g, pc, rc, sa, la, tp = load_fixture()
    rng = np.random.default_rng(3)
    sa_la0 = np.concatenate([[0], np.cumsum(sa["nla"])]).astype(np.int64)
    la_tp0 = np.concatenate([[0], np.cumsum(la["ntp"])]).astype(np.int64) * 2

    def batch(keys, tag):
        n = len(keys)
        ins = np.zeros(n, dtype=dentist_amd.INSERTION_REC_DTYPE)
        bases, ids, ov, las, trace = [], [], [], [], []
        for i, k in enumerate(keys):
            ins[i]["start_contig"], ins[i]["start_part"], ins[i]["end_contig"], ins[i]["end_part"] = k, 2, k + 1, 1
            L = int(rng.integers(1, 40))
            ins[i]["seq_len"], ins[i]["noverlaps"], ins[i]["nread_ids"] = L, 1 + (i & 1), 1 + i % 3
            bases.append(rng.integers(0, 4, L).astype(np.uint8))
            ids.append(np.asarray([tag * 1000 + k * 10 + j for j in range(1 + i % 3)], dtype=np.uint32))
            for j in range(1 + (i & 1)):
                s = int(rng.integers(0, len(sa)))
                ov.append(sa[s:s + 1])
                las.append(la[sa_la0[s]:sa_la0[s + 1]])
                trace.append(tp[la_tp0[sa_la0[s]]:la_tp0[sa_la0[s + 1]]])
        path = str(tmp_path / f"batch.{tag}.db")
        cat = lambda xs, like: np.concatenate(xs) if xs else like[:0]
        dentist_amd.insertiondb_write(path, ins, cat(bases, np.zeros(0, np.uint8)), cat(ids, np.zeros(0, np.uint32)),
                                      cat(ov, sa), cat(las, la), cat(trace, tp))
        return path, [(int(k), tag) for k in keys]

    files = [batch([1, 3, 5], 1), batch([4, 2, 3], 2), batch([3, 5, 6], 3), batch([], 4)]   # batch 2 is unsorted
    out = str(tmp_path / "insertions.db")
    assert dentist_amd.insertiondb_merge([f[0] for f in files], out) == 9
    m = dentist_amd.insertiondb_read(out)
    assert m["insertions"]["start_contig"].tolist() == [1, 2, 3, 3, 3, 4, 5, 5, 6]
    # every merged insertion carries its own payload: look each one up in its source file
    src = {f[0]: dentist_amd.insertiondb_read(f[0]) for f in files}
    got_tags = [int(m["read_ids"][o]) // 1000 for o in np.concatenate([[0], np.cumsum(m["insertions"]["nread_ids"])])[:-1]]
    assert got_tags == [1, 2, 1, 2, 3, 2, 1, 3, 3]          # ties in file order

    def payloads(d):
        ins = d["insertions"]
        bo = np.concatenate([[0], np.cumsum(ins["seq_len"])])
        io = np.concatenate([[0], np.cumsum(ins["nread_ids"])])
        so = np.concatenate([[0], np.cumsum(ins["noverlaps"])])
        lo = np.concatenate([[0], np.cumsum(d["seeded"]["nla"])])
        to = np.concatenate([[0], np.cumsum(d["las"]["ntp"])]) * 2
        res = {}
        for i in range(len(ins)):
            l0, l1 = lo[so[i]], lo[so[i + 1]]
            key = (int(ins[i]["start_contig"]), int(d["read_ids"][io[i]]) // 1000)
            res[key] = (ins[i].tobytes(), d["bases"][bo[i]:bo[i + 1]].tobytes(), d["read_ids"][io[i]:io[i + 1]].tobytes(),
                        d["seeded"][so[i]:so[i + 1]].tobytes(), d["las"][l0:l1].tobytes(), d["trace"][to[l0]:to[l1]].tobytes())
        return res
    want = {}
    for d in src.values():
        want.update(payloads(d))
    assert payloads(m) == want and len(want) == 9
    # no inputs: an empty container; a missing file: an error, nothing written over the old result
    assert dentist_amd.insertiondb_merge([], str(tmp_path / "empty.db")) == 0
    assert len(dentist_amd.insertiondb_read(str(tmp_path / "empty.db"))["insertions"]) == 0
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.insertiondb_merge([files[0][0], str(tmp_path / "nope.db")], out)
