"""The graph step of `dentist output` (common/scaffold.d) and the writer for ANY assembly graph: the reference's own unit
vectors (normalizeUnkownJoins scaffold.d:453-620, linearWalk :900-1019, scaffoldStarts :1297-1350, StringUniqifier
output.d:1068-1078) on the oracle restatement (oracle/output.py) AND on the product (dh_scaffold_graph_probe exposes
the graph code of dh_output.cpp); then product == oracle, byte for byte, on hand-made assemblies with anti-parallel
joins, contig-skipping joins, extensions and a cyclic scaffold, and on random assemblies under the three join policies
and the three `--only` settings.  CPU only."""
import ctypes
import json
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from oracle import output as oo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scaffold_graph.json")
V = json.load(open(GOLD))


def probe(n, joins, normalize=False, start=None, first=None, cap=64):
    """dh_scaffold_graph_probe with 1-based contig ids on both sides."""
    L = dentist_amd.lib()
    j = np.asarray([[a - 1, p, b - 1, q] for a, p, b, q in joins], dtype=np.int32).reshape(-1, 4)
    edges, starts, walk = np.zeros((cap, 4), np.int32), np.zeros((cap, 2), np.int32), np.zeros((cap, 4), np.int32)
    ne, ns, wl, cyc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    st = np.asarray([start[0] - 1, start[1]], np.int32) if start is not None else None
    fi = np.asarray([first[0] - 1, first[1], first[2] - 1, first[3]], np.int32) if first is not None else None
    L.dh_scaffold_graph_probe.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
    rc = L.dh_scaffold_graph_probe(n, j.ctypes.data if len(j) else None, len(j), int(normalize), edges.ctypes.data, ctypes.byref(ne),
                                   starts.ctypes.data, ctypes.byref(ns), st.ctypes.data if st is not None else None,
                                   fi.ctypes.data if fi is not None else None, walk.ctypes.data, ctypes.byref(wl), ctypes.byref(cyc), cap)
    assert rc == 0, dentist_amd.lib().dh_last_error()
    one = lambda a, k: [tuple(int(x) + (1 if i % 2 == 0 else 0) for i, x in enumerate(r)) for r in a[:k]]   # noqa: E731
    return one(edges, ne.value), one(starts, ns.value), one(walk, wl.value), bool(cyc.value)


def oracle_graph(n, joins):
    g = oo.Graph(n)
    for c in range(n):
        g.edges[((c, oo.BEGIN), (c, oo.END))] = ["contig", 0, []]
    for a, p, b, q in joins:
        g.edges[oo.edge_key((a - 1, p), (b - 1, q))] = ["plain"]
    return g


def canon(edges):
    return sorted(tuple(e) if (e[0], e[1]) <= (e[2], e[3]) else (e[2], e[3], e[0], e[1]) for e in edges)


def test_normalize_unknown_joins_vectors_of_the_reference():
    for case in V["normalize"]:
        want = canon(case["expect"])
        g = oo.normalize_unknown_joins(oracle_graph(case["n"], case["joins"]))
        assert canon([(a[0] + 1, a[1], b[0] + 1, b[1]) for a, b in g.edges]) == want
        edges, _, _, _ = probe(case["n"], case["joins"], normalize=True)
        assert canon(edges) == want


def walk_start(walk):
    """getWalkStart (scaffold.d:944): the node of the first join that does not connect it to the second."""
    a, b = (walk[0][0], walk[0][1]), (walk[0][2], walk[0][3])
    nxt = {(walk[1][0], walk[1][1]), (walk[1][2], walk[1][3])}
    return a if b in nxt else b


def directed(walk, start):
    out, at = [], start
    for e in walk:
        a, b = (e[0], e[1]), (e[2], e[3])
        to = b if a == at else a
        out.append((at[0], at[1], to[0], to[1]))
        at = to
    return out


def test_linear_walk_vectors_of_the_reference():
    g1 = V["walk_graph1"]
    og = oracle_graph(g1["n"], g1["joins"])
    for walk in g1["walks"]:
        for w in (walk, walk[::-1]):
            s = walk_start(w)
            want = directed(w, s)
            ow, cyc = oo.linear_walk(og, (s[0] - 1, s[1]))
            assert not cyc and directed([(a[0] + 1, a[1], b[0] + 1, b[1]) for a, b in ow], s) == want
            _, _, pw, pc = probe(g1["n"], g1["joins"], start=s)
            assert not pc and pw == want
    g2 = V["walk_graph2"]
    og = oracle_graph(g2["n"], g2["joins"])
    for w in (g2["walk"], g2["walk"][::-1]):
        s = walk_start(w)
        want = directed(w, s)
        first = w[0]
        ow, cyc = oo.linear_walk(og, (s[0] - 1, s[1]), oo.edge_key((first[0] - 1, first[1]), (first[2] - 1, first[3])))
        assert cyc and directed([(a[0] + 1, a[1], b[0] + 1, b[1]) for a, b in ow], s) == want
        _, _, pw, pc = probe(g2["n"], g2["joins"], start=s, first=first)
        assert pc and pw == want


def test_scaffold_starts_vectors_of_the_reference():
    for g in (V["walk_graph1"], V["walk_graph2"]):
        want = [tuple(x) for x in g["starts"]]
        assert [(c + 1, p) for c, p in oo.scaffold_starts(oracle_graph(g["n"], g["joins"]))] == want
        assert probe(g["n"], g["joins"])[1] == want


def test_string_uniqifier_vectors_of_the_reference(tmp_path):
    u = oo.StringUniqifier()
    for key, label, want in V["uniqifier"]:
        assert u(key, label) == want
    # the product: four single-contig scaffolds whose headers share an id
    contigs = sim.SeqDb.from_list([np.zeros(5, np.uint8)] * 4)
    fa = str(tmp_path / "u.fasta")
    dentist_amd.output_assembly(fa, contigs, [0, 1, 2, 3], ["A", "A", "A", "B"], [0, 0, 0, 0],
                                np.zeros(0, dtype=dentist_amd.INSERTION_DTYPE), np.zeros(0, np.uint8), line_width=0)
    assert [l for l in open(fa).read().split("\n") if l.startswith(">")] == [">A\tscaffold-1", ">A-1\tscaffold-2", ">A-2\tscaffold-3", ">B\tscaffold-4"]


# ------------------------------------------------------------------------------------------------ writer: product == oracle
def make_rec(entries):
    rec = np.zeros(len(entries), dtype=dentist_amd.INSERTION_DTYPE)
    for i, e in enumerate(entries):
        for k, v in e.items():
            rec[i][k] = v
    return rec


def both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy, only, line_width=0, min_ext=100, agp_mode="dazzler", tag="x"):
    names = ["read%d/%d" % (i, i * 7) for i in range(64)]
    fa, bed, agp = (str(tmp_path / (tag + n)) for n in (".fasta", ".bed", ".agp"))
    db = sim.SeqDb.from_list(contigs)
    off = np.asarray([0] + list(np.cumsum([len(x) for x in ids])), dtype=np.int64)
    flat = np.asarray([x for l in ids for x in l], dtype=np.int32)
    kw = dict(agp_dazzler=agp_mode == "dazzler", agp_skip_read_ids=agp_mode == "skip", read_names=names if agp_mode == "names" else None)
    dropped = dentist_amd.output_assembly(fa, db, sof, headers, gaps, rec, bases, read_ids=(flat, off), bed_path=bed, agp_path=agp,
                                          join_policy=["scaffoldGaps", "scaffolds", "contigs"][policy],
                                          only={1: "spanning", 2: "extending", 3: "both"}[only], min_extension_length=min_ext,
                                          line_width=line_width, tool="t", input_assembly="a.dam", **kw)
    efa, ebed, eagp, edropped = oo.write_assembly(contigs, sof, headers, gaps, rec, bases, read_ids=ids, policy=policy, only=only,
                                                  min_extension_length=min_ext, line_width=line_width, bed=True, read_names=names,
                                                  agp=dict(dazzler=agp_mode == "dazzler", skip_read_ids=agp_mode == "skip", tool="t",
                                                           input_assembly="a.dam"))
    assert open(fa).read() == efa
    assert open(bed).read() == ebed
    assert open(agp).read() == eagp
    assert dropped == edropped
    return efa, ebed, eagp, dropped


def hand_made():
    """Five contigs in two scaffolds (1-3 | 4-5): an anti-parallel end-end join 1 -> 2 in place of the first n run, a
    begin-begin join 2 -> 5 between the scaffolds that skips two contigs, a front extension of contig 1, a back extension
    of contig 4 (complement overlap), and one insertion that failed."""
    rng = np.random.default_rng(11)
    contigs = [rng.integers(0, 4, n).astype(np.uint8) for n in (300, 260, 240, 280, 220)]
    cons = [rng.integers(0, 4, n).astype(np.uint8) for n in (180, 200, 150, 170, 90)]
    off = np.cumsum([0] + [len(c) for c in cons])
    E = lambda i, **kw: dict(cons_off=int(off[i]), cons_len=len(cons[i]), ref_read_id=i, **kw)   # noqa: E731
    rec = make_rec([
        E(0, contig_left=0, contig_right=1, join=2, left_aepos=290, right_abpos=255, ins_begin=40, ins_end=120, comp=0),
        E(1, contig_left=1, contig_right=4, join=1, left_aepos=12, right_abpos=9, ins_begin=30, ins_end=160, comp=1),
        E(2, contig_left=0, contig_right=-1, join=1 | 4, left_aepos=7, right_abpos=-1, ins_begin=0, ins_end=110, comp=0),
        E(3, contig_left=3, contig_right=-1, join=4, left_aepos=270, right_abpos=-1, ins_begin=45, ins_end=170, comp=1),
        E(4, contig_left=2, contig_right=3, join=0, status=4),
    ])
    ids = [[3, 9, 1], [2], [5, 4], [8, 0, 6, 7], [1]]
    return contigs, [0, 0, 0, 1, 1], ["scafA extra\tmore", "scafB"], [25, 35, 0, 15, 0], rec, np.concatenate(cons), ids, cons


def test_hand_made_assembly_with_every_kind_of_join(tmp_path):
    contigs, sof, headers, gaps, rec, bases, ids, cons = hand_made()
    t = lambda a: sim.decode(a)   # noqa: E731
    rc = lambda a: sim.decode(sim.revcomp(a))   # noqa: E731
    # policy contigs, both kinds: one scaffold  3' <- ... walk from the smallest end node
    fa, bed, agp, dropped = both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy=2, only=3, tag="all")
    assert dropped == 0
    # components: contig 3 alone (its n run to contig 2 went where the anti-parallel join took contig 2's end);
    # the walk (1, pre) -> ext -> contig 1 -> [1 end | 2 end] -> contig 2 reversed -> [2 begin | 5 begin] -> contig 5
    # reversed? no: after two anti-parallel joins the strand is forward again -> contig 5 forward; contig 4 with its back
    # extension hangs on the n run before contig 5
    recs = dict(x.split("\n", 1) for x in fa.strip().split(">")[1:])
    first = recs["scafA extra\tscaffold-1"].strip()
    ext1 = t(cons[2][0:110]).upper()
    ins01 = t(cons[0][40:120]).upper()
    ins14 = rc(cons[1])[30:160]   # frame of the complement overlap on contig 2 ...
    walk = ext1 + t(contigs[0][7:290]) + ins01 + rc(contigs[1][12:255])
    assert first.startswith(walk)
    # ... entered from contig 2 while the strand is flipped: the insertion is written as the stored consensus reads
    assert first[len(walk):len(walk) + 130] == sim.decode(sim.revcomp(sim.encode(ins14))).upper()
    assert first.endswith(t(contigs[4][9:]))
    # spanning only: the extensions are gone, contig 1 starts at its first base
    fa2, _, _, _ = both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy=2, only=1, tag="span")
    assert t(contigs[0][0:290]) + ins01 in fa2.replace("\n", "")
    # the default policy keeps (c, end) -> (c + 1, begin) joins on n runs only: the join between the scaffolds goes, and so
    # does the anti-parallel one although it sits on an n run of scaffold A (scaffold.d:662-683)
    fa3, _, _, d3 = both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy=0, only=1, tag="gaps")
    assert d3 == 2 and t(contigs[0]) + "n" * 25 + t(contigs[1]) + "n" * 35 + t(contigs[2]) in fa3.replace("\n", "")
    for mode in ("names", "skip"):
        both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy=1, only=3, line_width=60, agp_mode=mode, tag=mode)


def test_cyclic_scaffold(tmp_path):
    rng = np.random.default_rng(12)
    contigs = [rng.integers(0, 4, n).astype(np.uint8) for n in (150, 170)]
    cons = [rng.integers(0, 4, 80).astype(np.uint8), rng.integers(0, 4, 90).astype(np.uint8)]
    rec = make_rec([
        dict(contig_left=0, contig_right=1, join=0, left_aepos=140, right_abpos=10, ins_begin=10, ins_end=60, comp=0, cons_off=0, cons_len=80),
        dict(contig_left=0, contig_right=1, join=1 | 2, left_aepos=5, right_abpos=160, ins_begin=20, ins_end=70, comp=0, cons_off=80, cons_len=90),
    ])
    fa, _, _, _ = both(tmp_path, contigs, [0, 1], ["c1", "c2"], [0, 0], rec, np.concatenate(cons), [[0], [1]], policy=2, only=1, tag="cyc")
    assert fa.startswith(">c1\tscaffold-1\tisCyclic\n")
    body = fa.split("\n", 1)[1].replace("\n", "")
    assert body == sim.decode(contigs[0][5:140]) + sim.decode(cons[0][10:60]).upper() + sim.decode(contigs[1][10:160]) + sim.decode(cons[1][20:70]).upper()


def test_random_assemblies_product_equals_oracle(tmp_path):
    rng = np.random.default_rng(2026)
    tried = 0
    for it in range(120):
        n = int(rng.integers(2, 9))
        contigs = [rng.integers(0, 4, int(rng.integers(120, 400))).astype(np.uint8) for _ in range(n)]
        sof, s = [], 0
        for c in range(n):
            sof.append(s)
            if rng.random() < 0.4:
                s += 1
        headers = ["s%d" % (i % 3) for i in range(s + 1)]   # duplicate ids on purpose
        gaps = [int(rng.integers(0, 40)) for _ in range(n)]
        # random joins on free contig ends (every end at most once), random extensions on some free ends
        free = [(c, p) for c in range(n) for p in (oo.BEGIN, oo.END)]
        rng.shuffle(free)
        entries, cons = [], []
        while len(free) >= 2 and rng.random() < 0.75:
            a = free.pop()
            k = next((i for i, b in enumerate(free) if b[0] != a[0]), None)
            if k is None:
                break
            b = free.pop(k)
            (c0, p0), (c1, p1) = sorted([a, b])
            cl = int(rng.integers(60, 200))
            x, y = sorted(int(v) for v in rng.integers(0, cl + 1, 2))
            l0, l1 = len(contigs[c0]), len(contigs[c1])
            entries.append(dict(contig_left=c0, contig_right=c1, join=(1 if p0 == oo.BEGIN else 0) | (2 if p1 == oo.END else 0),
                                left_aepos=int(rng.integers(0, 50)) if p0 == oo.BEGIN else l0 - int(rng.integers(0, 50)),
                                right_abpos=int(rng.integers(0, 50)) if p1 == oo.BEGIN else l1 - int(rng.integers(0, 50)),
                                ins_begin=x, ins_end=y, comp=int(rng.integers(0, 2)), cons_len=cl, status=0 if rng.random() < 0.9 else 3))
            cons.append(rng.integers(0, 4, cl).astype(np.uint8))
        for (c, p) in list(free):
            if rng.random() < 0.4:
                free.remove((c, p))
                cl = int(rng.integers(60, 300))
                x = int(rng.integers(0, cl + 1))
                front = p == oo.BEGIN
                entries.append(dict(contig_left=c, contig_right=-1, join=4 | (1 if front else 0),
                                    left_aepos=int(rng.integers(0, 50)) if front else len(contigs[c]) - int(rng.integers(0, 50)), right_abpos=-1,
                                    ins_begin=0 if front else x, ins_end=x if front else cl, comp=int(rng.integers(0, 2)), cons_len=cl))
                cons.append(rng.integers(0, 4, cl).astype(np.uint8))
        order = sorted(range(len(entries)), key=lambda i: (entries[i]["contig_left"], entries[i]["join"] & 1 == 0, entries[i]["contig_right"] if entries[i]["contig_right"] >= 0 else 10 ** 6))
        entries = [entries[i] for i in order]
        cons = [cons[i] for i in order]
        at = 0
        for e, c in zip(entries, cons):
            e["cons_off"], e["ref_read_id"] = at, int(rng.integers(0, 64))
            at += len(c)
        rec = make_rec(entries)
        bases = np.concatenate(cons) if cons else np.zeros(0, np.uint8)
        ids = [sorted(set(int(v) for v in rng.integers(0, 64, int(rng.integers(1, 6))))) for _ in entries]
        for policy in (0, 1, 2):
            only = int(rng.integers(1, 4))
            both(tmp_path, contigs, sof, headers, gaps, rec, bases, ids, policy=policy, only=only,
                 line_width=[0, 50, 70][it % 3], min_ext=[0, 100][it % 2], agp_mode=["dazzler", "names", "skip"][it % 3], tag="r%d_%d" % (it, policy))
            tried += 1
    assert tried == 360
