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
(pileups.d:1124-1590) is restated below (find_cyclic_subgraphs = Paton's cycle base, util/math.d:2362-2480, pinned by
its unittest :2488-2535; the BubbleResolver's graph surgery); the re-mapping of the skipping reads onto the
intermediate contigs (getReadAlignmentsOnContigs, :1316-1385: the external aligner) is a callback.

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


# ---------------------------------------------------------------- resolveBubbles, pileups.d:1100-1590
def incident_edges(edges, nodes):
    """IncidentEdgesCache (math.d:1080-1130): per node the incident edges in edge order (a self loop once)."""
    idx = {n: i for i, n in enumerate(nodes)}
    inc = [[] for _ in nodes]
    for e in edges:
        inc[idx[e["start"]]].append(e)
        if e["end"] != e["start"]:
            inc[idx[e["end"]]].append(e)
    return inc


def find_cyclic_subgraphs(nodes, edges, inc=None):
    """Paton's cycle base as util/math.d:2380-2480 walks it: roots in node order, a LIFO of discovered nodes, the
    incident edges of a node in edge order.  Returns cycles as lists of node indices."""
    idx = {n: i for i, n in enumerate(nodes)}
    if inc is None:
        inc = incident_edges(edges, nodes)
    n = len(nodes)
    used = [set() for _ in range(n)]
    parent = [-1] * n
    cycles = []
    for root in range(n):
        if parent[root] >= 0:
            continue
        parent[root] = root
        used[root].add(root)
        stack = [root]
        while stack:
            cur = stack.pop()
            for e in inc[cur]:
                nb = idx[e["end"] if e["start"] == nodes[cur] else e["start"]]
                if not used[nb]:
                    parent[nb] = cur
                    used[nb].add(cur)
                    stack.append(nb)
                elif nb == cur:
                    cycles.append([cur])
                elif nb not in used[cur]:
                    cyc = [nb, cur]
                    p = parent[cur]
                    while p not in used[nb]:
                        cyc.append(p)
                        p = parent[p]
                    cyc.append(p)
                    cycles.append(cyc)
                    used[nb].add(cur)
    return cycles


def is_extension_join(e):
    return is_front_ext_join(e) or is_back_ext_join(e)


def _node_matches(node, sa):
    """contigNodeMatchesReadAlignment (pileups.d:1493-1510)."""
    return node[0] == sa[0]["a_id"] and ((node[1] == BEGIN and sa[1] == FRONT) or (node[1] == END and sa[1] == BACK))


def collect_fixed_simple_bubbles(same_read, skipped_path):
    """pileups.d:1414-1491: the read alignments of one skipping read after the re-mapping, valid iff they walk the
    skipped path in order."""
    ras = collect_read_alignments(same_read)
    if not ras:
        return []
    path = skipped_path[::-1] if skipped_path[0][0] != ras[0][0][0]["a_id"] else skipped_path
    flat = [sa for ra in ras for sa in ra]
    at = next((i for i, sa in enumerate(flat) if _node_matches(path[0], sa)), None)
    if at is None or len(path) > len(flat) - at:
        return []
    if any(not _node_matches(nd, sa) for nd, sa in zip(path, flat[at:])):
        return []
    return ras


def resolve_bubbles(edges, num_contigs, remap, max_bubble_size=8, max_iterations=4):
    """BubbleResolver.run (pileups.d:1124-1315).  remap(skipping pile-up, intermediate contig ids) -> alignment chains of
    the skipping reads on the intermediate contigs (disabled unless they cover their contig completely)."""
    nodes = nodes_of(num_contigs)
    idx = {n: i for i, n in enumerate(nodes)}
    for _ in range(max_iterations):
        inc = incident_edges(edges, nodes)
        deg = [sum(1 for e in ie if not is_extension_join(e)) for ie in inc]
        find = lambda a, b: next((e for e in edges if _key(e) == _key(edge(a, b))), None)  # noqa: E731

        def simple(cyc):
            esc = [nodes[i] for i in cyc if deg[i] >= 3]
            if len(esc) != 2 or any(deg[i] < 2 for i in cyc):
                return False
            sk = find(esc[0], esc[1])
            return sk is not None and bool(sk["types"] & T_PILEUP)
        bubbles = [c for c in find_cyclic_subgraphs(nodes, edges, inc) if len(c) <= max_bubble_size and simple(c)]
        if not bubbles:
            break
        for cyc in bubbles:
            esc = [nodes[i] for i in cyc if deg[i] >= 3]
            sk = find(esc[0], esc[1])
            if not sk["types"] & T_PILEUP:
                continue   # resolved through another bubble of this iteration
            pile = sk["ras"]
            inter = sorted({nodes[i][0] for i in cyc if deg[i] == 2})
            new = remap(pile, inter)
            augmented = [sa[0] for ra in pile for sa in ra] + list(new)
            i0, i1 = cyc.index(idx[sk["start"]]), cyc.index(idx[sk["end"]])
            walk = lambda a, b: [cyc[(a + x) % len(cyc)] for x in range((b - a) % len(cyc) + 1)]  # noqa: E731
            path = walk(i0, i1)
            if len(path) == 2:
                path = walk(i1, i0)
            assert len(path) > 2, "skipped path is too short"
            path = [nodes[i] for i in path]
            al = sorted([a for a in augmented], key=lambda a: a["b_id"])
            al = [a for a in al if not a["disabled"]]
            joins, i = [], 0
            while i < len(al):
                j = i
                while j < len(al) and al[j]["b_id"] == al[i]["b_id"]:
                    j += 1
                for ra in collect_fixed_simple_bubbles(al[i:j], path):
                    if ra_is_valid(ra):
                        joins.append(make_join(ra_in_order(ra)))
                i = j
            sk["types"] &= ~T_PILEUP
            sk["ras"] = []
            edges = bulk_add(edges, joins, merge_joins)
        edges = remove_none_joins(edges)
    return edges


def build(num_contigs, alignments, input_gaps, min_spanning_reads=3, best_pile_up_margin=3.0,
          existing_gap_bonus=6.0, merge_extensions=True, remap=None, max_bubble_size=8, max_bubble_iterations=4):
    """pileups.d:173-208.  input_gaps: (begin contig id, end contig id); remap: the re-mapping callback of
    resolve_bubbles (None: resolveBubbles is skipped, cyclic subgraphs are left to discardAmbiguousJoins)."""
    joins = collect_scaffold_joins(alignments)
    joins += [edge((b, END), (e, BEGIN), T_INPUTGAP) for b, e in input_gaps]
    sc = build_scaffold(num_contigs, joins)
    if remap is not None:
        sc = resolve_bubbles(sc, num_contigs, remap, max_bubble_size, max_bubble_iterations)
    sc = discard_ambiguous_joins(sc, num_contigs, best_pile_up_margin, existing_gap_bonus)
    sc = enforce_min_spanning_reads(sc, min_spanning_reads)
    sc = remove_input_gaps(sc)
    if merge_extensions:
        sc = merge_extensions_with_gaps(sc, num_contigs)
    return collect_pile_ups(sc)
