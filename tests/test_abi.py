"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import glob
import json
import os
import re

import numpy as np
import pytest

import dentist_amd
from dentist_amd import _lib

from test_oracle_golden import GOLD, dentist_flags, parse_ladump  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    L = dentist_amd.lib()
    decl = declared_symbols()
    assert decl, "no declarations found"
    for name in sorted(decl):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert decl == set(_lib.SYMBOLS), "binding list and header disagree"
    assert L.dh_abi_version() == 4


def test_process_opts_layout_matches_the_abi_version():
    """dh_process_opts grew by two int32 fields with ABI 4 (max_partners, min_relative_score_ppm): the binding's struct is
    the header's (16 x int32 = 64 bytes), the defaults fill the new fields, and a zero there is refused."""
    assert ctypes.sizeof(_lib.ProcessOpts) == 64
    hdr = open(os.path.join(ROOT, "include", "dentist_hip.h")).read()
    body = hdr[hdr.index("int32_t tspace_map;"):hdr.index("} dh_process_opts;")]
    fields = re.findall(r"^\s*int32_t\s+(\w+);", body, flags=re.M)
    assert fields == [n for n, _ in _lib.ProcessOpts._fields_], fields
    o = dentist_amd.default_process_opts()
    assert o.min_relative_score_ppm == 1000000 and o.max_partners == 0


def o_algo_default():
    return dentist_amd.default_align_opts().algo


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.AlignOpts) == 68 and o_algo_default() == 0
    assert _lib.LA_DTYPE.itemsize == 48
    o = dentist_amd.default_align_opts()
    assert (o.k, o.hmin, o.band_shift, o.tspace, o.min_len, o.pen) == (14, 35, 6, 100, 500, 6)
    assert o.width <= 62 and o.algo == 0


def test_no_silent_cpu_fallback_without_gpu():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dentist_amd.DhError) as ei:
        dentist_amd.Context(0)
    assert ei.value.code == -2  # DH_ENODEV


def test_product_las_codec_matches_oracle_bytes(tmp_path):
    """Host-only I/O of the product library: same bytes as the oracle's codec on the golden dump."""
    import json
    from test_oracle_golden import parse_ladump, GOLD
    from oracle import pyoracle as oz
    las, trace, _ = parse_ladump(json.load(open(os.path.join(GOLD, "las_dump.json")))["dump"])
    for ts in (100, 126):
        p1, p2 = str(tmp_path / f"a{ts}.las"), str(tmp_path / f"b{ts}.las")
        oz.las_write(p1, las, trace, ts)
        dentist_amd.las_write(p2, las, trace, ts)
        assert open(p1, "rb").read() == open(p2, "rb").read()
        l2, t2, ts2 = dentist_amd.las_read(p1)
        assert ts2 == ts and np.array_equal(t2, trace)
        for f in ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread"):
            assert np.array_equal(l2[f], las[f])
    open(p1, "wb").write(open(p1, "rb").read()[:-1])
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.las_read(p1)


def test_output_fasta_writer_unclosed_gaps_and_wrapping(tmp_path):
    """`dentist output` subset (output.d:743-925) on the host: header rule, n-runs for open gaps,
    upper-cased insertion, reverse-complemented consensus, 50-column wrapping (no GPU needed)."""
    from dentist_amd import sim
    from dentist_amd._lib import INSERTION_DTYPE
    rng = np.random.default_rng(5)
    c = [rng.integers(0, 4, n).astype(np.uint8) for n in (120, 80, 60, 40)]
    contigs = sim.SeqDb.from_list(c)
    cons = rng.integers(0, 4, 30).astype(np.uint8)
    rec = np.zeros(2, dtype=INSERTION_DTYPE)
    # gap 0|1 closed with the reverse complement of cons[5:25]; gap 1|2 skipped (status != 0)
    rec[0] = (0, 0, 5, 0, 7, 0, 0, 110, 4, 5, 25, 1, 30, 0, 0, 0, 0, 1, 0)
    rec[1] = (1, 4, 5, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0)
    path = str(tmp_path / "o.fasta")
    dentist_amd.output_fasta(path, contigs, [0, 0, 0, 1], ["chrA\tfoo", "chrB"], [33, 17, 0], rec, cons)
    txt = open(path).read().split(">")[1:]
    assert txt[0].split("\n")[0] == "chrA\tscaffold-1" and txt[1].split("\n")[0] == "chrB\tscaffold-4"
    body = txt[0].split("\n")[1:]
    assert all(len(l) == 50 for l in body[:-2]) and 0 < len(body[-2]) <= 50 and body[-1] == ""
    s0 = "".join(body)
    ins = sim.decode(sim.revcomp(cons)[5:25]).upper()
    want = sim.decode(c[0][:110]) + ins + sim.decode(c[1][4:]) + "n" * 17 + sim.decode(c[2])
    assert s0 == want
    assert "".join(txt[1].split("\n")[1:]) == sim.decode(c[3])


def test_las_merge_of_block_files(tmp_path):
    """LAmerge (Snakefile:1173-1185): per-block .las files merge into one file in LAsort order,
    traces intact (host only)."""
    from dentist_amd._lib import LA_DTYPE
    rng = np.random.default_rng(3)

    def block(breads):
        las = np.zeros(len(breads), dtype=LA_DTYPE)
        tr = []
        for i, b in enumerate(breads):
            n = int(rng.integers(1, 5))
            las[i]["aread"], las[i]["bread"] = int(rng.integers(0, 4)), b
            las[i]["abpos"], las[i]["aepos"] = 100 * i, 100 * i + 100 * n
            las[i]["bbpos"], las[i]["bepos"] = 0, 100 * n
            las[i]["tlen"], las[i]["toff"] = 2 * n, len(tr)
            tr += [int(x) for x in rng.integers(0, 200, 2 * n)]
            las[i]["diffs"] = sum(tr[-2 * n::2])
        order = np.lexsort((las["abpos"], las["bread"], las["aread"]))
        return las[order], np.asarray(tr, dtype=np.uint16)

    files, every = [], []
    for k, br in enumerate(([0, 1, 2, 3], [4, 5], [6, 7, 8])):
        las, tr = block(br)
        p = str(tmp_path / f"ref.reads.{k + 1}.las")
        dentist_amd.las_write(p, las, tr, 100)
        files.append(p)
        every += [(int(l["aread"]), int(l["bread"]), tr[l["toff"]:l["toff"] + l["tlen"]].tolist()) for l in las]
    out = str(tmp_path / "ref.reads.las")
    dentist_amd.las_merge(files, out)
    las, tr, ts = dentist_amd.las_read(out)
    assert ts == 100 and len(las) == len(every)
    keys = [(int(l["aread"]), int(l["bread"])) for l in las]
    assert keys == sorted(keys)
    got = sorted((int(l["aread"]), int(l["bread"]), tr[l["toff"]:l["toff"] + l["tlen"]].tolist()) for l in las)
    assert got == sorted(every)


def _golden_la():
    g = json.load(open(os.path.join(GOLD, "trace_cases.json")))
    la = g["la"]
    rec = np.zeros(1, dtype=dentist_amd.LA_DTYPE)
    tr = np.asarray(la["tp"], dtype=np.uint16).reshape(-1)
    rec[0]["abpos"], rec[0]["aepos"], rec[0]["bbpos"], rec[0]["bepos"] = la["abpos"], la["aepos"], la["bbpos"], la["bepos"]
    rec[0]["diffs"], rec[0]["tlen"], rec[0]["toff"] = la["diffs"], len(tr), 0
    return g, rec[0], tr, la["tspace"]


def test_product_trace_translation_matches_the_reference_vectors():
    """base.d:883-944 / :244-264 (translateTracePoint) applied to the PRODUCT's arithmetic through the
    C ABI (dh_translate_trace_point is what dh_crop_pileups cuts reads with), not only to the oracle."""
    g, la, tr, ts = _golden_la()
    for apos, ea, eb in g["floor_cases"]:
        assert dentist_amd.translate_trace_point(la, tr, ts, apos, "floor") == (ea, eb)
    for p1, m1, p2, m2 in g["equal_pairs"]:
        assert dentist_amd.translate_trace_point(la, tr, ts, p1, m1) == dentist_amd.translate_trace_point(la, tr, ts, p2, m2)


def test_product_crop_to_trace_point_cases_of_the_reference():
    """base.d:993-1129 (cropToTracePoint): a front crop keeps [begin, tp], a back crop [tp, end]; the
    chain is disabled when nothing is left; positions outside the LA are an error."""
    g, la, tr, ts = _golden_la()

    def crop(seed, pos, mode):
        a, b = dentist_amd.translate_trace_point(la, tr, ts, pos, mode)
        empty = (a == la["abpos"] or b == la["bbpos"]) if seed == "front" else (a == la["aepos"] or b == la["bepos"])
        return empty, a, b
    for seed, pos, mode, disabled, ea, eb in g["crop_cases"]:
        empty, a, b = crop(seed, pos, mode)
        assert empty == disabled
        if not disabled:
            assert (a, b) == (ea, eb)
    for seed, p1, m1, p2, m2 in g["equal_crop_pairs"]:
        assert crop(seed, p1, m1) == crop(seed, p2, m2)
    for pos in g["crop_throws"]:
        with pytest.raises(dentist_amd.DhError):
            dentist_amd.translate_trace_point(la, tr, ts, pos, "floor")


def test_product_las_filter_roundtrip_of_the_reference(tmp_path):
    """dazzler.d:3901-3988: read test.las, keep the even ids, write, read back -> the six expected
    records (product codec on both legs)."""
    g = json.load(open(os.path.join(GOLD, "las_dump.json")))
    las, trace, _ = parse_ladump(g["dump"])
    src = str(tmp_path / "test.las")
    dentist_amd.las_write(src, las, trace, 100)
    las1, trace1, ts = dentist_amd.las_read(src)
    keep = las1[::2].copy()
    dst = str(tmp_path / "test-filtered.las")
    dentist_amd.las_write(dst, keep, trace1, ts)
    las2, trace2, ts2 = dentist_amd.las_read(dst)
    assert ts2 == 100 and len(las2) == len(g["filtered_even"])
    for la, exp in zip(las2, g["filtered_even"]):
        assert [la["aread"] + 1, la["abpos"], la["aepos"]] == exp["a"]
        assert [la["bread"] + 1, la["bbpos"], la["bepos"]] == exp["b"]
        assert sorted(dentist_flags(int(la["flags"]))) == sorted(exp["flags"])
        assert trace2[la["toff"]:la["toff"] + la["tlen"]].reshape(-1, 2).tolist() == exp["tp"]


def test_pileups_of_general_joins_are_validated():
    """dh_pileups_create_joins: nodes ordered, contig0 < contig1, seeds 0 / 1, an extension has no second contig; the
    plain creator's pile-ups read back as (c, back, c + 1, front); the sharded glue refuses general joins loudly."""
    tri = [[(0, 0, 1)], [(1, 2, -1)]]
    p = dentist_amd.Pileups.from_joins([(0, 1, 3, 1), (2, 0, -1, 0)], tri)
    assert len(p) == 2 and p.get_join(0) == (0, 1, 3, 1) and p.get_join(1) == (2, 0, -1, 0)
    assert p.get(1)[1].tolist() == [[1, 2, -1]]
    q = dentist_amd.Pileups.from_triples([4, 7], [np.asarray([(0, 0, 1)]), np.asarray([(2, 3, 4)])])
    assert q.get_join(0) == (4, 1, 5, 0) and q.get_join(1) == (7, 1, 8, 0)
    for bad in ([(3, 1, 0, 1), (4, 0, -1, 0)],          # contig0 > contig1
                [(2, 0, -1, 0), (0, 1, 3, 1)],          # not ordered by their nodes
                [(0, 2, 3, 1), (4, 0, -1, 0)],          # seed out of range
                [(0, 1, 0, 0), (4, 0, -1, 0)]):         # a contig joined with itself
        with pytest.raises(dentist_amd.DhError):
            dentist_amd.Pileups.from_joins(bad, tri)
    las = np.zeros(5, dtype=dentist_amd.LA_DTYPE)
    with pytest.raises(dentist_amd.DhError, match="general joins"):
        __import__("dentist_amd._lib", fromlist=["x"]).shard_pack_candidates(p, las)
