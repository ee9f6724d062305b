"""Oracle-side restatement of DENTIST's `collect` (spanning reads only) and `process` call
sequence, driving the C oracle.  TEST INFRASTRUCTURE ONLY (never imported by dentist_amd/).

Follows source/dentist/commands/processPileUps/package.d:283-374 (crop -> pile-up alignment ->
filter -> QV -> reference read -> consensus -> flank re-alignment -> insertion), the cropping
arithmetic of processPileUps/cropper.d:446-550 and the splice coordinates of
common/insertions.d:110-146.  Pure-Python loops: small cases only.
"""
import numpy as np

from dentist_amd.sim import SeqDb, revcomp
from . import pyoracle as oz

MIN_ANCHOR = 500       # commandline.d:2036 minAnchorLength
ALLOWANCE_MAP = 100    # proper-alignment-allowance = trace spacing (commandline.d:2331)
TS_PILE = 126
WAVE_WIDTH = 30  # live diagonals of a wave: the product default (dh_default_align_opts), two alignments per wavefront
MAX_PILE = 60          # reads kept per pile-up (one wavefront tracks <= 64 aligned regions)
MAX_INS_ERR_PPM = 100000  # commandline.d:1997 maxInsertionError 0.10


def collect_spanning(las, trace, contigs, reads, allowance=ALLOWANCE_MAP, min_anchor=MIN_ANCHOR, min_reads=3,
                     max_reads=MAX_PILE):
    """Gap g lies between contig g and g+1.  A read spans it when it has an LA reaching the end of
    contig g and an LA starting at the begin of contig g+1, same orientation, in read order.  A read
    enters a pile-up once: with its qualifying pair of the longest anchors (ties: lowest indices)."""
    by_read = {}
    for i, la in enumerate(las):
        by_read.setdefault(int(la["bread"]), []).append(i)
    piles = {}
    for r, idxs in sorted(by_read.items()):
        best = {}
        for iL in idxs:
            L = las[iL]
            if L["flags"] & 0x20 or int(L["aread"]) + 1 >= contigs.n:   # 0x20: dropped by the collect filters
                continue
            cl = contigs.length(int(L["aread"]))
            if L["aepos"] + allowance < cl or L["aepos"] - L["abpos"] < min_anchor:
                continue
            for iR in idxs:
                R = las[iR]
                if R["flags"] & 0x20 or R["aread"] != L["aread"] + 1 or (R["flags"] & 1) != (L["flags"] & 1):
                    continue
                if R["abpos"] > allowance or R["aepos"] - R["abpos"] < min_anchor:
                    continue
                if R["bbpos"] + allowance < L["bepos"] - allowance:  # right part must follow the left part
                    continue
                anchors = int(L["aepos"] - L["abpos"]) + int(R["aepos"] - R["abpos"])
                g = int(L["aread"])
                if g not in best or anchors > best[g][0]:
                    best[g] = (anchors, iL, iR)
        for g, (_, iL, iR) in best.items():
            piles.setdefault(g, []).append((r, iL, iR))
    out = {}
    for g, v in piles.items():
        if len(v) < min_reads:   # min-reads-per-pile-up (commandline.d:2125-2187)
            continue
        if len(v) > max_reads:   # keep the reads whose anchoring LAs have the lowest error rate
            def err(e):
                L, R = las[e[1]], las[e[2]]
                ln = int(L["aepos"] - L["abpos"]) + int(R["aepos"] - R["abpos"])
                return (int(L["diffs"]) + int(R["diffs"])) * 1000000 // max(ln, 1)
            keep = sorted(range(len(v)), key=lambda x: (err(v[x]), x))[:max_reads]
            v = [v[x] for x in sorted(keep)]
        out[g] = v
    return out


def ceil_to(x, m):
    return -(-x // m) * m


def chain_members(las, i):
    """Records of the alignment chain that starts at record i: NEXT without START continues it (dazzler.d:1728-1758)."""
    j = i + 1
    while j < len(las) and (las[j]["flags"] & 0x8) and not (las[j]["flags"] & 0x4) and las[j]["aread"] == las[j - 1]["aread"] \
            and las[j]["bread"] == las[j - 1]["bread"] and (las[j]["flags"] & 1) == (las[j - 1]["flags"] & 1):
        j += 1
    return list(range(i, j))


def region_of(las, i):
    """to!(ReferenceRegion, "contigA") of a chain (common/package.d:228-241): the union of its members' A intervals."""
    iv = sorted((int(las[x]["abpos"]), int(las[x]["aepos"])) for x in chain_members(las, i))
    out = []
    for b, e in iv:
        if out and b <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([b, e])
    return out


def intersect_regions(regs):
    cur = regs[0]
    for r in regs[1:]:
        nxt = []
        for b0, e0 in cur:
            for b1, e1 in r:
                b, e = max(b0, b1), min(e0, e1)
                if b < e:
                    nxt.append([b, e])
        cur = nxt
    return cur


def subtract_mask(reg, mask):
    """Region - mask (util/region.d opBinary!"-"): mask = sorted disjoint (begin, end) pairs."""
    out = []
    for b, e in reg:
        for mb, me in mask:
            if me <= b:
                continue
            if mb >= e:
                break
            if mb > b:
                out.append([b, mb])
            b = max(b, me)
        if b < e:
            out.append([b, e])
    return out


def common_trace_point(regions, contig_len, ts, seed_front, mask=None):
    """getCommonTracePoint, cropper.d:446-500.  regions: one list of [begin, end) per alignment chain (a plain
    (begin, end) pair counts as one interval); mask: the contig's repeat mask -- the region outside it is tried
    first, then the common region itself."""
    regs = [[list(r)] if not isinstance(r[0], (list, tuple)) else [list(x) for x in r] for r in regions]
    common = intersect_regions(regs)
    for reg in ([subtract_mask(common, mask)] if mask else []) + [common]:
        if not reg:
            continue
        lo, hi = reg[0][0], reg[-1][1]
        tp_min, tp_sup = ceil_to(lo, ts), ceil_to(hi, ts)
        cands = list(range(tp_min, tp_sup, ts)) + ([contig_len] if tp_sup > contig_len else [])
        if seed_front:
            cands = cands[::-1]
        for c in cands:
            if any(b <= c < e for b, e in reg) or c == hi:
                return c
    return -1


def translate_floor(la, tr, apos, ts):
    import ctypes
    a, b = ctypes.c_int32(), ctypes.c_int32()
    t = np.ascontiguousarray(tr, dtype=np.uint16)
    oz.lib().oz_translate_trace_point_a(int(la["abpos"]), int(la["aepos"]), int(la["bbpos"]), ts, t.ctypes.data,
                                        len(t) // 2, int(apos), 0, ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


FRONT, BACK = 0, 1   # AlignmentLocationSeed


def join_of(g):
    """A pile-up's join as (contig0, seed0, contig1, seed1): an int g is the gap between contig g and g + 1 in the
    same orientation, (g, BACK, g + 1, FRONT); contig1 = -1 is an extension pile-up (one flank)."""
    return (int(g), BACK, int(g) + 1, FRONT) if not isinstance(g, (tuple, list)) else tuple(int(x) for x in g)


def crop_pile(entries, las, trace, contigs, reads, g, ts_map=100, mask=None):
    """cropPileUp (cropper.d:113-175) for any join: returns (crop0, crop1, SeqDb of cropped reads in read orientation,
    read ids, kinds).  An entry is (read, LA on flank 0, LA on flank 1); an extension-type read alignment merged into
    a gap (scaffold.d:789-816) has None / -1 in place of the alignment it lacks.  Per alignment getCroppingSlice
    (cropper.d:339-348, 503-550): a back-seeded one keeps [crop point, end), a front-seeded one [0, crop point) of the
    read as the alignment sees it, mirrored for a complement alignment; the slices of an entry are intersected.
    Support patches per alignment (getSingleReadPatch / getReadPatches, cropper.d:351-378): to the read's front when
    (contig seed == front) == complement, else to its back.  kind: 0 both flanks, 1 flank 0 only, 2 flank 1 only,
    | 4 / 8: the alignment on flank 0 / 1 is a complement one.  mask: {contig: [(begin, end), ...]} repeat mask."""
    has = lambda i: i is not None and i >= 0   # noqa: E731
    c0, s0, c1, s1 = join_of(g)
    flanks = [(c0, s0)] + ([(c1, s1)] if c1 >= 0 else [])
    crops = []
    for f, (c, sd) in enumerate(flanks):
        regs = [region_of(las, e[1 + f]) for e in entries if has(e[1 + f])]
        crops.append(common_trace_point(regs, contigs.length(c), ts_map, sd == FRONT, mask=(mask or {}).get(c)) if regs else -1)
    if any(c < 0 for c in crops):
        return None
    # fetchSupportPatches (cropper.d:224-262): if less than minAnchorLength of a flank remains after
    # cropping, the missing piece of the contig is glued to every cropped read
    patches = []
    for (c, sd), pos in zip(flanks, crops):
        cs = contigs.seq(c)
        if sd == FRONT:
            patches.append(cs[pos:MIN_ANCHOR] if pos < MIN_ANCHOR else cs[0:0])
        else:
            patches.append(cs[max(0, len(cs) - MIN_ANCHOR):pos] if len(cs) - pos < MIN_ANCHOR else cs[0:0])
    seqs, ids, kinds = [], [], []
    for e in entries:
        r = e[0]
        rl = reads.length(r)
        b0, b1 = 0, rl
        pre, post = patches[0][0:0], patches[0][0:0]
        kind = 0 if has(e[1]) and has(e[2]) else (1 if has(e[1]) else 2)
        for f, ((c, sd), pos) in enumerate(zip(flanks, crops)):
            i = e[1 + f]
            if not has(i):
                continue
            # AlignmentChain.translateTracePoint (base.d:866-880): the FIRST member that covers the position translates it
            M = las[next(x for x in chain_members(las, i) if las[x]["abpos"] <= pos <= las[x]["aepos"])]
            _, b = translate_floor(M, trace[M["toff"]:M["toff"] + M["tlen"]], pos, ts_map)
            lo, hi = (0, b) if sd == FRONT else (b, rl)
            comp = bool(las[i]["flags"] & 1)
            if comp:   # complement -> swap and mirror (cropper.d:533-538)
                lo, hi = rl - hi, rl - lo
                kind |= 4 << f
            b0, b1 = max(b0, lo), min(b1, hi)
            patch = revcomp(patches[f]) if comp else patches[f]
            if (sd == FRONT) == comp:
                pre = patch
            else:
                post = patch
        if b1 - b0 < 14:
            continue
        seqs.append(np.concatenate([pre, reads.seq(r)[b0:b1], post]).astype(np.uint8))
        ids.append(r)
        kinds.append(kind)
    return crops[0], (crops[1] if len(crops) > 1 else -1), SeqDb.from_list(seqs), ids, kinds


def stage_width(algo, band=64):
    """DH-2 (algo 1): the band (64 rows, or 32 when dh_process_opts.width = 32); DH-1: the product's default wave width."""
    return (32 if band == 32 else 64) if algo == 1 else WAVE_WIDTH


def pile_opts(algo=0, nreads=0, band=64):
    # skip_self = 2: every unordered pair aligned once, both records emitted (what daligner does);
    # record slots / candidates per (read, strand) grow with the pile-up (a read overlaps every other read)
    max_la = 64 if nreads <= 60 else (128 if nreads <= 124 else 256)
    return oz.default_opts(tspace=TS_PILE, min_len=500, skip_self=2, max_la=max_la, max_cand=min(256, 2 * max_la),
                           width=stage_width(algo, band), algo=algo)


def filter_pile_las(las, pile, max_err_ppm=300000, allowance=TS_PILE, proper=True, min_rel_score=1.0):
    """computeQVs' alignment funnel (processPileUps/package.d:474-516): filterLocalAlignments
    (averageErrorRate <= maxAlignmentError) -> chainLocalAlignments -> filterPileUpAlignments
    (properAlignmentAllowance, forceFlat; dazzler.d:4066-4094).  Dropped LAs get DISABLED (0x20)."""
    las = las.copy()
    for la in las:
        al = int(la["aepos"] - la["abpos"])
        if int(la["diffs"]) * 1000000 > max_err_ppm * al:
            la["flags"] |= 0x20
    las = chain_pile_las(las, min_rel_score=min_rel_score)
    if proper:
        las = filter_proper(las, pile, allowance)
    return las


def filter_proper(las, pile, allowance=TS_PILE):
    """filterPileUpAlignments (dazzler.d:4043-4094); runs AFTER the tile QVs (package.d:492-512)."""
    las = las.copy()
    for la in las:
        if la["flags"] & 0x20:
            continue
        if not oz.valid_pileup_alignment(la, pile.length(int(la["aread"])), pile.length(int(la["bread"])), allowance):
            la["flags"] |= 0x20
    return las


def chain_pile_las(las, max_indel=1000, max_gap=10000, max_rel_overlap=0.3, min_rel_score=1.0, min_score=TS_PILE):
    """chainLocalAlignments / buildAlignmentChains (common/alignments/chaining.d:122-334) with the defaults of
    commandline.d:1819, 1982, 2014, 2153, 2165-2173, per (A, B) pair:
      * the pair's enabled LAs are split into the connected components of the undirected chainability relation (:182);
      * per component a shortest-path problem rates the chains (node bonus = mean length, edge penalty = indel + gap/10,
        :227-233; relaxations over the LAs ordered by (abpos, bbpos, index), a topological order);
      * per component the end nodes within effectiveMinScore of the component's best chain are taken best first
        (:236-266): a node already on a taken chain is no end node; a chain that runs into nodes of a better chain is an
        ALTERNATE chain and is composed of its whole path (:269-285) -- the shared LAs are written once per chain;
      * the chains scoring >= max(minScore, minRelativeScore * best of the pair) are accepted (:305-312).
    Flags: first LA of a chain START (+ BEST unless alternate, dazzler.d:2063-2068), the others NEXT; LAs on no accepted
    chain DISABLED.  Returns a NEW array: further occurrences of shared LAs are inserted behind their first occurrence.
    Ties (two end nodes / two predecessors with equal distance): the lower position in the (abpos, bbpos, index) order
    first -- the reference's unstable sort (:240-245) and its depth-first topological order (util/graphalgo.d:926-960)
    are not properties of the data."""
    las = las.copy()
    groups = {}
    for i, la in enumerate(las):
        if la["flags"] & 0x20:
            continue
        groups.setdefault((int(la["aread"]), int(la["bread"])), []).append(i)

    def score(x):
        return (int(x["aepos"] - x["abpos"]) + int(x["bepos"] - x["bbpos"])) // 2

    def chainable(x, y):
        if (x["flags"] & 1) != (y["flags"] & 1):
            return False
        ga, gb = int(y["abpos"]) - int(x["aepos"]), int(y["bbpos"]) - int(x["bepos"])
        if not (x["abpos"] < y["abpos"] and x["bbpos"] < y["bbpos"]):
            return False
        if abs(ga - gb) > max_indel or max(abs(ga), abs(gb)) > max_gap:
            return False
        mla = min(int(x["aepos"] - x["abpos"]), int(y["aepos"] - y["abpos"]))
        mlb = min(int(x["bepos"] - x["bbpos"]), int(y["bepos"] - y["bbpos"]))
        return max(0, -ga) <= max_rel_overlap * mla and max(0, -gb) <= max_rel_overlap * mlb

    def chain_score(x, y):
        ga, gb = int(y["abpos"]) - int(x["aepos"]), int(y["bbpos"]) - int(x["bepos"])
        return abs(ga - gb) + max(abs(ga), abs(gb)) // 10 - score(y)

    CMASK = 0x4 | 0x8 | 0x10
    dups = []   # (record index, flags)
    for idxs in groups.values():
        order = sorted(idxs, key=lambda i: (int(las[i]["abpos"]), int(las[i]["bbpos"]), i))   # a topological order
        n = len(order)
        dist = [-score(las[i]) for i in order]
        pred = [-1] * n
        comp = list(range(n))
        for u in range(n):
            for v in range(u + 1, n):
                if chainable(las[order[u]], las[order[v]]):
                    d = dist[u] + chain_score(las[order[u]], las[order[v]])
                    if dist[v] > d:
                        dist[v], pred[v] = d, u
                    cu, cv = comp[u], comp[v]
                    if cu != cv:
                        comp = [cu if c == cv else c for c in comp]
        members = {}
        for v in range(n):
            members.setdefault(comp[v], []).append(v)
        sel = []   # (end node, alternate, score)
        forbidden = set()
        for c in sorted(members, key=lambda c: min(order[v] for v in members[c])):
            ends = sorted(members[c], key=lambda v: (dist[v], v))
            cthr = int(max(min_score, min_rel_score * -dist[ends[0]]))
            for e in ends:
                if e in forbidden or -dist[e] < cthr:
                    continue
                alt, v = False, e
                while v >= 0:
                    alt = alt or v in forbidden
                    forbidden.add(v)
                    v = pred[v]
                sel.append((e, alt, -dist[e]))
        best = max((sc for _, _, sc in sel), default=0)
        thr = int(max(min_score, min_rel_score * best))
        seen = set()
        for e, alt, sc in sel:
            if sc < thr:
                continue
            path, v = [], e
            while v >= 0:
                path.append(v)
                v = pred[v]
            path.reverse()
            for k, v in enumerate(path):
                f = (int(las[order[v]]["flags"]) & ~CMASK) | ((0x4 | (0 if alt else 0x10)) if k == 0 else 0x8)
                if v not in seen:
                    seen.add(v)
                    las[order[v]]["flags"] = f
                else:
                    dups.append((order[v], f))
        for v in range(n):
            if v not in seen:
                las[order[v]]["flags"] |= 0x20
    if dups:
        dups.sort(key=lambda d: d[0])
        at = [d[0] + 1 for d in dups]
        rows = las[[d[0] for d in dups]].copy()
        rows["flags"] = [d[1] for d in dups]
        las = np.insert(las, at, rows)
    return las


def process_pile(entries, las, trace, contigs, reads, g, rounds=3, nthreads=1, flank_window=20000, dust=True, algo=0, mask=None,
                 max_partners=0, min_rel_score=1.0, band=64):
    """One pile-up through the `process` sequence; returns a dict describing the insertion.  g: the left contig of a
    plain gap, or any join (contig0, seed0, contig1, seed1) -- see join_of."""
    res = {"gap": g, "status": "ok", "nreads": len(entries)}
    c0, s0, c1, s1 = join_of(g)
    flanks = [(c0, s0)] + ([(c1, s1)] if c1 >= 0 else [])
    if c1 == c0:
        res["status"] = "unsupported join"
        return res
    crop = crop_pile(entries, las, trace, contigs, reads, g, mask=mask)
    if crop is None:
        res["status"] = "no common trace point"
        return res
    cropL, cropR, pile, ids, kinds = crop
    res.update(cropL=cropL, cropR=cropR, pile=pile, read_ids=ids, kinds=kinds)
    if pile.n < 3:
        res["status"] = "pile too small"
        return res
    o = pile_opts(algo, pile.n, band)
    if dust:   # DBdust pileup.db; daligner ... -mdust (package.d:476-482)
        pile = oz.with_dust(pile)
        res["pile"] = pile
    # allowed reference reads = the reads that span the gap (selectAllowedReferenceReadIds, package.d:461-472);
    # coverage = max(their number, 4 if pile >= 4) (package.d:498-503)
    # (an extension pile-up has one flank: every read has its alignment there)
    allowed = np.asarray([1 if (k & 3) == (0 if c1 >= 0 else 1) else 0 for k in kinds], dtype=np.uint8)
    # Everything downstream reads the overlaps of the allowed reads only, so (DH-2) records with another read as A are not
    # made and pairs of two such reads not aligned (oz_db.pflags bit 0); max_partners bounds the B side as well (bit 1): the
    # first max_partners reads in the order allowed reads, then the others, each in pile-up order (dh_process_opts).
    if algo == 1 and allowed.any():
        order = [i for i in range(pile.n) if allowed[i]] + [i for i in range(pile.n) if not allowed[i]]
        partner = np.ones(pile.n, dtype=np.uint8)
        if max_partners > 0 and pile.n > max_partners:
            partner[:] = 0
            partner[order[:max_partners]] = 1
        fl = (allowed | (partner << 1)).astype(np.uint8)
        pile.pflags = np.ascontiguousarray(fl) if (fl != 3).any() else None
    plas, ptrace, _ = oz.align_db(pile, pile, o, nthreads=nthreads)
    pile.pflags = None
    # computeQVs (package.d:474-516): error filter -> chain -> DAScover/DASqv -> proper-overlap filter
    chained = filter_pile_las(plas, pile, proper=False, min_rel_score=min_rel_score)
    rlen = np.asarray([pile.length(i) for i in range(pile.n)], dtype=np.int32)
    cov = int(allowed.sum())
    if cov < 4 and pile.n >= 4:
        cov = 4
    qv = oz.tile_qv(chained, ptrace, rlen, TS_PILE, max(cov, 1))
    plas = filter_proper(chained, pile)
    res.update(pile_las=plas, pile_trace=ptrace, chained_las=chained)
    if not np.any((plas["flags"] & 0x20) == 0):
        res["status"] = "empty pileup alignment after filtering"
        return res
    order, badqv = oz.rank_reference_reads(qv, rlen, TS_PILE, allowed=allowed)
    if len(order) == 0:
        res["status"] = "pile too small"   # no valid reference read (package.d:335-343)
        return res
    ref_idx = int(order[0])
    res.update(qv=qv, order=order, ref_idx=ref_idx)
    cons = oz.consensus(pile.seq(ref_idx), pile, plas, ptrace, ref_idx, TS_PILE)
    for _ in range(1, rounds):
        tdb = SeqDb.from_list([cons])
        o2 = oz.default_opts(tspace=TS_PILE, min_len=500, max_la=4, max_cand=32, width=stage_width(algo, band), algo=algo)
        rl, rt, _ = oz.align_db(tdb, pile, o2, nthreads=nthreads)
        for la in rl:   # proper overlaps only
            if not oz.valid_pileup_alignment({**{f: la[f] for f in la.dtype.names}, "aread": -1},
                                             len(cons), pile.length(int(la["bread"])), TS_PILE):
                la["flags"] |= 0x20
        cons = oz.consensus(cons, pile, rl, rt, 0, TS_PILE)
    res["consensus"] = cons
    # flank re-alignment: daligner -A -s126 -l126 contigs consensus (commandline.d:2918-2935); one slice per flank:
    # the contig's tail for a back-seeded flank (starting on the contig's trace grid), its head for a front-seeded one
    fw = flank_window if flank_window > 0 else max(contigs.length(c) for c, _ in flanks)   # 0 = the whole contigs
    offs, slices = [], []
    for c, sd in flanks:
        cs = contigs.seq(c)
        wl = 0 if sd == FRONT else max(0, len(cs) - fw) // TS_PILE * TS_PILE
        offs.append(wl)
        slices.append(cs[:fw] if sd == FRONT else cs[wl:])
    fdb = SeqDb.from_list(slices)
    if dust:   # DBdust contigs.dam; daligner -A ... -mdust -mrep (package.d:631-667)
        fdb = oz.with_dust(fdb)
    o3 = oz.default_opts(tspace=TS_PILE, min_len=126, max_la=4, max_cand=32, width=stage_width(algo, band), algo=algo)
    fl_las, fl_tr, _ = oz.align_db(fdb, SeqDb.from_list([cons]), o3, nthreads=nthreads)
    res.update(flank_las=fl_las, flank_trace=fl_tr, flank_off=offs[0], flank_offs=offs)
    allow = TS_PILE
    refk = kinds[ref_idx] >> 2   # complement flags of the reference read's alignments: the consensus has its orientation
    ov = []
    for f, (c, sd) in enumerate(flanks):
        # an overlap whose complement flag differs from the reference read's on that contig is disabled
        # (package.d:669-690); exactly one proper insertion overlap per flank (:707-745)
        cand = [la for la in fl_las if la["aread"] == f and bool(la["flags"] & 1) == bool((refk >> f) & 1)]
        if sd == FRONT:
            cand = [la for la in cand if la["abpos"] <= allow and la["bepos"] + allow >= len(cons)]
        else:
            cand = [la for la in cand if la["aepos"] + allow >= len(slices[f]) and la["bbpos"] <= allow]
        ov.append(cand)
    if any(len(x) != 1 for x in ov):
        res["status"] = "consensus does not align uniquely to the flanks (%s)" % ",".join(str(len(x)) for x in ov)
        return res
    ov = [x[0] for x in ov]
    L = ov[0]
    if len(ov) == 2 and ((L["flags"] & 1) == (ov[1]["flags"] & 1)) != (s0 != s1):   # isParallel as the reference read (package.d:757-773)
        res["status"] = "flank orientation mismatch"
        return res
    for la in ov:
        if int(la["diffs"]) * 1000000 > MAX_INS_ERR_PPM * int(la["aepos"] - la["abpos"]):   # ensureHighQualityConsensus output.d:388-410
            res["status"] = "maxInsertionError"
            return res
    cseq = revcomp(cons) if (L["flags"] & 1) else cons
    # getCroppingPosition (insertions.d:110-146): contigA -- front seed: begin of the overlap, back seed: its end;
    # contigB likewise, here in the frame of the flank-0 overlap
    res["left_aepos"] = offs[0] + int(L["abpos"] if s0 == FRONT else L["aepos"])
    p0 = int(L["bbpos"] if s0 == FRONT else L["bepos"])
    if len(ov) == 2:
        R = ov[1]
        res["right_abpos"] = offs[1] + int(R["abpos"] if s1 == FRONT else R["aepos"])
        p1 = int(R["bbpos"] if s1 == FRONT else R["bepos"])
        if (R["flags"] & 1) != (L["flags"] & 1):
            p1 = len(cons) - p1
        b0, b1 = (p1, p0) if s0 == FRONT else (p0, p1)
    else:
        res["right_abpos"] = -1
        b0, b1 = (0, p0) if s0 == FRONT else (p0, len(cons))
    res.update(ins_begin=b0, ins_end=b1, comp=int(L["flags"] & 1))
    if b1 < b0:
        res["status"] = "negative insertion"
        return res
    res["insertion"] = cseq[b0:b1].copy()
    return res
