"""Oracle restatement of the alignment filters of `dentist collect`
(source/dentist/commands/collectPileUps/filter.d:122-356, order collectPileUps/package.d:130-141) on
alignment chains (base.d:306-421): records linked by START / NEXT (dazzler.d:1728-1758) are one unit --
first.begin .. last.end, totalDiffs / coveredBases, union of the members' A intervals minus the mask.
TEST INFRASTRUCTURE ONLY.  Plain Python loops: small cases only.
Predicates restated from the reference's own D code: averageErrorRate base.d:695-698, isProper
:537-556, isFullyContained :562-598 (unit cases :600-640), toInterval common/package.d:259-288,
AlignmentChain.opCmp base.d:766-777."""
import numpy as np

DISABLED = 0x20


def is_proper(la, alen, blen, allowance):
    begins = la["abpos"] <= allowance or la["bbpos"] <= allowance
    ends = la["aepos"] + allowance >= alen or la["bepos"] + allowance >= blen
    return bool(begins and ends)


def is_fully_contained(la, alen, blen):
    if la["bbpos"] > la["abpos"]:
        return False
    y = int(la["aepos"]) + blen - int(la["bepos"])
    return y < alen


def b_interval(la, blen):
    if la["flags"] & 1:
        return blen - int(la["bepos"]), blen - int(la["bbpos"])
    return int(la["bbpos"]), int(la["bepos"])


START, NEXT = 0x4, 0x8


def chain_ranges(las):
    """[(first record, end record)] of every chain: NEXT without START continues the chain of the record before it."""
    out, i, n = [], 0, len(las)
    while i < n:
        j = i + 1
        while j < n and (las[j]["flags"] & NEXT) and not (las[j]["flags"] & START) and las[j]["aread"] == las[j - 1]["aread"] \
                and las[j]["bread"] == las[j - 1]["bread"] and (las[j]["flags"] & 1) == (las[j - 1]["flags"] & 1):
            j += 1
        out.append((i, j))
        i = j
    return out


def chain_units(las):
    """One pseudo record per chain (first member's fields, end of the last, diffs summed) + coveredBases!"contigA"."""
    rng = chain_ranges(las)
    u = np.zeros(len(rng), dtype=las.dtype)
    cov = []
    for c, (i, j) in enumerate(rng):
        u[c] = las[i]
        u[c]["aepos"], u[c]["bepos"] = las[j - 1]["aepos"], las[j - 1]["bepos"]
        u[c]["diffs"] = int(las["diffs"][i:j].sum())
        if np.any(las["flags"][i:j] & DISABLED):
            u[c]["flags"] |= DISABLED
        cov.append(int((las["aepos"][i:j].astype(np.int64) - las["abpos"][i:j]).sum()))
    return u, cov, rng


def collect_filter(las, contig_off, read_off, max_align_err=0.30, allowance=100, min_anchor=500, repeat_mask=None):
    """The six filters with alignment chains as units; returns (records with DISABLED set on every member of a dropped
    chain, chains dropped per stage, read_used)."""
    las = las.copy()
    u, cov, rng = chain_units(las)
    if all(j - i == 1 for i, j in rng):
        return _filter_units(las, contig_off, read_off, max_align_err, allowance, min_anchor, repeat_mask)
    unm = []
    for i, j in rng:   # size of (union of the members' A intervals) - mask
        iv = sorted((int(las[x]["abpos"]), int(las[x]["aepos"])) for x in range(i, j))
        merged = []
        for b, e in iv:
            if merged and b <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([b, e])
        size = 0
        for b, e in merged:
            size += e - b
            if repeat_mask is not None:
                ptr, ivs = repeat_mask
                a = int(las[i]["aread"])
                for q in range(int(ptr[a]), int(ptr[a + 1])):
                    lo, hi = max(int(ivs[2 * q]), b), min(int(ivs[2 * q + 1]), e)
                    if hi > lo:
                        size -= hi - lo
        unm.append(size)
    fu, dropped, used = _filter_units(u, contig_off, read_off, max_align_err, allowance, min_anchor, repeat_mask, cov, unm)
    for c, (i, j) in enumerate(rng):
        if fu[c]["flags"] & DISABLED:
            las["flags"][i:j] |= DISABLED
    return las, dropped, used


def _filter_units(las, contig_off, read_off, max_align_err=0.30, allowance=100, min_anchor=500, repeat_mask=None, cov=None,
                  unmasked_of=None):
    las = las.copy()
    n = len(las)
    alen = lambda l: int(contig_off[l["aread"] + 1] - contig_off[l["aread"]])   # noqa: E731
    blen = lambda l: int(read_off[l["bread"] + 1] - read_off[l["bread"]])       # noqa: E731
    dropped = []

    def ndis():
        return int(((las["flags"] & DISABLED) != 0).sum())
    base = ndis()
    for x, l in enumerate(las):   # LQ: totalDiffs / coveredBases!"contigA"
        covered = cov[x] if cov is not None else int(l["aepos"] - l["abpos"])
        if not l["flags"] & DISABLED and int(l["diffs"]) * 1000000 > int(round(max_align_err * 1e6)) * covered:
            l["flags"] |= DISABLED
    dropped.append(ndis() - base); base = ndis()
    for l in las:   # Improper
        if not l["flags"] & DISABLED and not is_proper(l, alen(l), blen(l), allowance):
            l["flags"] |= DISABLED
    dropped.append(ndis() - base); base = ndis()
    for x, l in enumerate(las):   # WeaklyAnchored
        if l["flags"] & DISABLED:
            continue
        unmasked = int(l["aepos"] - l["abpos"])
        if unmasked_of is not None:
            unmasked = unmasked_of[x]
        elif repeat_mask is not None:
            ptr, iv = repeat_mask
            for j in range(int(ptr[l["aread"]]), int(ptr[l["aread"] + 1])):
                b, e = max(int(iv[2 * j]), int(l["abpos"])), min(int(iv[2 * j + 1]), int(l["aepos"]))
                if e > b:
                    unmasked -= e - b
        if unmasked <= min_anchor:
            l["flags"] |= DISABLED
    dropped.append(ndis() - base); base = ndis()
    order = sorted(range(n), key=lambda i: (int(las[i]["aread"]), int(las[i]["bread"]), int(las[i]["abpos"]),
                                            int(las[i]["bbpos"]), int(las[i]["aepos"]), int(las[i]["bepos"]), i))
    for x in range(n):   # Contained
        a1 = las[order[x]]
        if a1["flags"] & DISABLED:
            continue
        b1 = b_interval(a1, blen(a1))
        for y in range(x + 1, n):
            a2 = las[order[y]]
            if not (a2["aread"] == a1["aread"] and a1["abpos"] <= a2["abpos"] and a2["aepos"] <= a1["aepos"]):
                break
            b2 = b_interval(a2, blen(a2))
            if (a2["flags"] & 1) == (a1["flags"] & 1) and a2["bread"] == a1["bread"] and b1[0] <= b2[0] and b2[1] <= b1[1]:
                las[order[y]]["flags"] |= DISABLED
    dropped.append(ndis() - base); base = ndis()
    nreads = len(read_off) - 1
    used = np.ones(nreads, dtype=np.uint8)
    by_read = {}
    for i in range(n):
        by_read.setdefault(int(las[i]["bread"]), []).append(i)
    for r, idx in by_read.items():   # Ambiguous
        live = [i for i in idx if not las[i]["flags"] & DISABLED]
        amb = False
        for p in range(len(live)):
            for q in range(p + 1, len(live)):
                bp, bq = b_interval(las[live[p]], blen(las[live[p]])), b_interval(las[live[q]], blen(las[live[q]]))
                if bp[0] < bq[1] and bq[0] < bp[1]:
                    amb = True
        if amb:
            used[r] = 0
            for i in idx:
                las[i]["flags"] |= DISABLED
    dropped.append(ndis() - base); base = ndis()
    for r, idx in by_read.items():   # Redundant
        if any(not las[i]["flags"] & DISABLED and is_fully_contained(las[i], alen(las[i]), blen(las[i])) for i in idx):
            used[r] = 0
            for i in idx:
                las[i]["flags"] |= DISABLED
    dropped.append(ndis() - base)
    return las, np.asarray(dropped, dtype=np.int64), used
