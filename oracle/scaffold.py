"""TEST INFRASTRUCTURE -- CPU restatement of the scaffold-graph pile-up builder of `dentist collect`.

Only tests/ may import this module.  It restates, in plain Python, what the reference does in
  source/dentist/commands/collectPileUps/pileups.d:173-208   build
  source/dentist/commands/collectPileUps/pileups.d:435-444   collectPileUps
  source/dentist/commands/collectPileUps/pileups.d:626-677   mergeJoins / selectMeanest / collectScaffoldJoins / makeScaffoldJoin
  source/dentist/commands/collectPileUps/pileups.d:796-888   makeScaffoldJoin(inputGap) / collectReadAlignments
  source/dentist/commands/collectPileUps/pileups.d:1592-1657 discardAmbiguousJoins
  source/dentist/commands/collectPileUps/pileups.d:1754-1804 findCorrectGapJoin
  source/dentist/commands/collectPileUps/pileups.d:1807-1852 enforceMinSpanningReads / removeInputGaps
  source/dentist/common/scaffold.d:75-230, 237-356, 772-817  nodes, join predicates, buildScaffold, removeNoneJoins, mergeExtensionsWithGaps
  source/dentist/common/alignments/base.d:1964-2050, 2160-2330, 2680-2790  SeededAlignment, ReadAlignment, makeJoin, PileUp predicates
  source/dentist/util/math.d:380-545, 684-688, 1467-1480, 1581-1603  undirected edges, bulkAdd, filterEdges, mapEdges

Pinned by the reference's own unittests (tests/test_scaffold.py): collectReadAlignments cases 1-5
(pileups.d:897-1097), mergeExtensionsWithGaps (scaffold.d:819-876), discardAmbiguousJoins
(pileups.d:1659-1730) and the 22-read build() case (pileups.d:210-432).

Where the reference sorts with an unstable sort and then merges equal elements (bulkAdd, findCorrectGapJoin)
this restatement sorts stably, i.e. it fixes the order the reference leaves open.  resolveBubbles
(pileups.d:1124-1590) re-aligns reads with the external tools and is not restated: cyclic subgraphs
are left to discardAmbiguousJoins, which removes their forks.

An alignment chain is a dict: id, contigA (id, length), contigB (id, length), complement, disabled and
first/last local alignment coordinates: a_begin, a_end, b_begin, b_end (b on the oriented read).
"""
FRONT, BACK = 0, 1                    # AlignmentLocationSeed
PRE, BEGIN, END, POST = 0, 1, 2, 3    # ContigPart
T_PILEUP, T_INPUTGAP = 1, 2           # ScaffoldPayload.Type


def chain(id, a_id, a_len, b_id, b_len, complement, a_begin, a_end, b_begin, b_end, disabled=False):
    return dict(id=id, a_id=a_id, a_len=a_len, b_id=b_id, b_len=b_len, complement=bool(complement),
                a_begin=a_begin, a_end=a_end, b_begin=b_begin, b_end=b_end, disabled=disabled)


# ---------------------------------------------------------------- base.d:2018-2050
def is_front_extension(ac):
    return ac["b_begin"] > ac["a_begin"]


def is_back_extension(ac):
    return ac["b_len"] - ac["b_end"] > ac["a_len"] - ac["a_end"]


def seeded_from(ac):
    """SeededAlignment.from (base.d:2002-2014): (chain, seed) copies."""
    out = []
    if is_front_extension(ac):
        out.append((ac, FRONT))
    if is_back_extension(ac):
        out.append((ac, BACK))
    return out


# ---------------------------------------------------------------- ReadAlignment, base.d:2160-2330
def ra_is_extension(ra):
    return len(ra) == 1


def ra_is_gap(ra):
    return len(ra) == 2 and ra[0][0]["a_id"] != ra[1][0]["a_id"] and ra[0][0]["b_id"] == ra[1][0]["b_id"]


def ra_is_valid(ra):
    return ra_is_extension(ra) != ra_is_gap(ra)


def ra_in_order(ra):
    if ra_is_gap(ra) and not ra[0][0]["a_id"] < ra[1][0]["a_id"]:
        return [ra[1], ra[0]]
    return ra


def ra_is_parallel(ra):
    return ra_is_gap(ra) and ra[0][1] != ra[1][1] and ra[0][0]["complement"] == ra[1][0]["complement"]


def ra_is_anti_parallel(ra):
    return ra_is_gap(ra) and ra[0][1] == ra[1][1] and ra[0][0]["complement"] != ra[1][0]["complement"]


# ---------------------------------------------------------------- pileups.d:821-888
def _begin_rel_b(ac):
    return ac["b_len"] - ac["b_end"] if ac["complement"] else ac["b_begin"]


def _end_rel_b(ac):
    return ac["b_len"] - ac["b_begin"] if ac["complement"] else ac["b_end"]


def _seed_rel_b(sa):
    return -sa[1] if sa[0]["complement"] else sa[1]


def collect_read_alignments(same_read_alignments):
    seeded = [sa for ac in same_read_alignments for sa in seeded_from(ac)]
    seeded.sort(key=lambda sa: (_begin_rel_b(sa[0]), _end_rel_b(sa[0]), _seed_rel_b(sa)))
    if not seeded:
        return []
    for a, b in zip(seeded, seeded[1:]):
        share = _end_rel_b(a[0]) > _begin_rel_b(b[0])
        parts_of_one = a[0] is b[0] and a[1] != b[1]
        if share and not parts_of_one:
            return []  # no region of the read may be used twice
    start_with_extension = _begin_rel_b(seeded[0][0]) > 0
    s0 = 1 if start_with_extension else 0
    ras = [seeded[i:min(i + 2, len(seeded))] for i in range(s0, len(seeded), 2)]
    if start_with_extension:
        ras = [seeded[0:1]] + ras
    if any(not ra_is_valid(ra) for ra in ras):
        return []
    return ras


# ---------------------------------------------------------------- graph (math.d, scaffold.d)
def edge(n0, n1, types=0, ras=None):
    """Undirected edge, start <= end (math.d:385-398); nodes are (contig id, part)."""
    if n1 < n0:
        n0, n1 = n1, n0
    return dict(start=n0, end=n1, types=types, ras=list(ras or []))


def _key(e):
    return (e["start"], e["end"])


def is_default(e):
    return e["start"][1] == BEGIN and e["end"][1] == END and e["start"][0] == e["end"][0]


def _real(part):
    return part in (BEGIN, END)


def is_gap(e):
    return e["start"][0] != e["end"][0] and _real(e["start"][1]) and _real(e["end"][1])


def is_front_ext_join(e):
    return e["start"][0] == e["end"][0] and e["start"][1] == PRE and e["end"][1] == BEGIN


def is_back_ext_join(e):
    return e["start"][0] == e["end"][0] and e["start"][1] == END and e["end"][1] == POST


def _payload_empty(e):
    return e["types"] == 0 and not e["ras"]


def remove_none_joins(edges):
    return [e for e in edges if is_default(e) or not _payload_empty(e)]


def bulk_add(edges, new, merge):
    """math.d:1467-1480 with a stable sort."""
    allE = sorted(edges + new, key=_key)
    out, i = [], 0
    while i < len(allE):
        j = i
        while j < len(allE) and _key(allE[j]) == _key(allE[i]):
            j += 1
        out.append(merge(allE[i:j]))
        i = j
    return out


def merge_joins(group):
    m = dict(group[0])
    if len(group) > 1:
        m["types"] = 0
        m["ras"] = []
        for g in group:
            m["types"] |= g["types"]
            m["ras"] = m["ras"] + g["ras"]
    return m


def select_meanest(group):
    return min(group, key=lambda g: bin(g["types"]).count("1"))  # first of the minimal ones


def build_scaffold(num_contigs, raw_joins, merge=merge_joins):
    """scaffold.d:237-244: default edges of contigs 1..n, raw joins merged, empty joins removed."""
    edges = [edge((c, BEGIN), (c, END)) for c in range(1, num_contigs + 1)]
    return remove_none_joins(bulk_add(edges, list(raw_joins), merge))


def nodes_of(num_contigs):
    return [(c, p) for c in range(1, num_contigs + 1) for p in (PRE, BEGIN, END, POST)]


def make_join(ra):
    """base.d:2680-2722."""
    if ra_is_gap(ra):
        part = lambda s: BEGIN if s == FRONT else END  # noqa: E731
        return edge((ra[0][0]["a_id"], part(ra[0][1])), (ra[1][0]["a_id"], part(ra[1][1])), T_PILEUP, [ra])
    c = ra[0][0]["a_id"]
    if ra[0][1] == FRONT:
        return edge((c, PRE), (c, BEGIN), T_PILEUP, [ra])
    return edge((c, END), (c, POST), T_PILEUP, [ra])


def collect_scaffold_joins(alignments):
    """pileups.d:650-667."""
    al = sorted(alignments, key=lambda a: a["b_id"])
    al = [a for a in al if not a["disabled"]]
    joins, i = [], 0
    while i < len(al):
        j = i
        while j < len(al) and al[j]["b_id"] == al[i]["b_id"]:
            j += 1
        for ra in collect_read_alignments(al[i:j]):
            if ra_is_valid(ra):
                joins.append(make_join(ra_in_order(ra)))
        i = j
    return joins


def find_correct_gap_join(gap_joins, margin, bonus):
    """pileups.d:1754-1804; len(gap_joins) = none."""
    vals = [(len(g["ras"]) * (bonus if g["types"] & T_INPUTGAP else 1.0), i) for i, g in enumerate(gap_joins)]
    vals.sort(key=lambda v: -v[0])
    if vals[1][0] * margin < vals[0][0]:
        return vals[0][1]
    return len(gap_joins)


def discard_ambiguous_joins(edges, num_contigs, margin, bonus):
    """pileups.d:1592-1657."""
    acc = []
    for node in nodes_of(num_contigs):
        inc = [e for e in edges if e["start"] == node or e["end"] == node]
        if _real(node[1]) and len(inc) > 2:
            gj = [e for e in inc if is_gap(e) and e["types"] & T_PILEUP]
            if len(gj) > 1:
                k = find_correct_gap_join(gj, margin, bonus)
                if k < len(gj):
                    gj = gj[:k] + gj[k + 1:]
                acc += gj
    removed = []
    for e in acc:
        r = dict(e)
        r["types"] = e["types"] & ~T_PILEUP
        r["ras"] = []
        removed.append(r)
    return remove_none_joins(bulk_add(edges, removed, select_meanest))


def enforce_min_spanning_reads(edges, min_spanning):
    out = []
    for e in edges:
        e = dict(e)
        if e["types"] & T_PILEUP and is_gap(e) and len(e["ras"]) < min_spanning:
            e["types"] &= ~T_PILEUP
            e["ras"] = []
        out.append(e)
    return remove_none_joins(sorted(out, key=_key))


def remove_input_gaps(edges):
    out = []
    for e in edges:
        e = dict(e)
        e["types"] &= ~T_INPUTGAP
        out.append(e)
    return remove_none_joins(sorted(out, key=_key))


def merge_extensions_with_gaps(edges, num_contigs, merge_payloads=None):
    """scaffold.d:789-816.  Edges emptied on the way stay in the graph (and count towards the degree)
    until the final removeNoneJoins, as in the reference."""
    edges = [dict(e) for e in edges]
    if merge_payloads is None:
        def merge_payloads(a, b):
            return a["types"] | b["types"], a["ras"] + b["ras"]
    for node in nodes_of(num_contigs):
        inc = [e for e in edges if e["start"] == node or e["end"] == node]
        assert len(inc) <= 3, "node degree must be <= 3"
        if _real(node[1]) and len(inc) == 3:
            nd = [e for e in inc if not is_default(e)]
            assert len(nd) == 2
            other = lambda e: e["end"] if e["start"] == node else e["start"]  # noqa: E731
            gi = 0 if _real(other(nd[0])[1]) else 1
            g, x = nd[gi], nd[1 - gi]
            g["types"], g["ras"] = merge_payloads(g, x)
            x["types"], x["ras"] = 0, []
    return remove_none_joins(edges)


# ---------------------------------------------------------------- PileUp predicates, base.d:2734-2790
def pile_is_extension(p):
    if p and len(p[0]) == 1 and p[0][0][1] == FRONT:
        return all(len(ra) == 1 and ra[0][1] == FRONT for ra in p)
    if p and len(p[0]) == 1 and p[0][0][1] == BACK:
        return all(len(ra) == 1 and ra[0][1] == BACK for ra in p)
    return False


def pile_is_gap(p):
    return any(ra_is_gap(ra) for ra in p)


def pile_is_valid(p):
    return pile_is_extension(p) != pile_is_gap(p)


def collect_pile_ups(edges):
    """pileups.d:435-444; keeps the join with the pile-up."""
    return [(e, e["ras"]) for e in edges if e["types"] & T_PILEUP and e["ras"] and pile_is_valid(e["ras"])]


def build(num_contigs, alignments, input_gaps, min_spanning_reads=3, best_pile_up_margin=3.0,
          existing_gap_bonus=6.0, merge_extensions=True):
    """pileups.d:173-208 without resolveBubbles.  input_gaps: (begin contig id, end contig id)."""
    joins = collect_scaffold_joins(alignments)
    joins += [edge((b, END), (e, BEGIN), T_INPUTGAP) for b, e in input_gaps]
    sc = build_scaffold(num_contigs, joins)
    sc = discard_ambiguous_joins(sc, num_contigs, best_pile_up_margin, existing_gap_bonus)
    sc = enforce_min_spanning_reads(sc, min_spanning_reads)
    sc = remove_input_gaps(sc)
    if merge_extensions:
        sc = merge_extensions_with_gaps(sc, num_contigs)
    return collect_pile_ups(sc)
