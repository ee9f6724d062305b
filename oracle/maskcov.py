"""TEST INFRASTRUCTURE -- CPU restatement of DENTIST's alignment-coverage mask.

Only tests/ may import this module.  Restates commands/maskRepetitiveRegions.d:238-430
(BadAlignmentCoverageAssessor.opCall, coverageZone, Masker) and :432-580 (CoverageChangeRange: events
sorted by (contig, position), all events of one position folded into one change).  Pinned by the
reference's unittests: coverageChanges (:591-631) and the assessor with limits (3, 5) (:394-410), both
on the 33 intervals of :299-333 (tests/test_maskcov.py)."""
import math


def coverage_changes(alignment_intervals, contig_intervals):
    """(contig, position, current coverage, new coverage) per distinct event position."""
    if not alignment_intervals:
        return []
    ev = []
    for c, b, e in alignment_intervals:
        ev.append((c, b, 1))
        ev.append((c, e, -1))
    for c, b, e in contig_intervals:
        ev.append((c, 0, 0))
        ev.append((c, e - b, 0))
    ev.sort()
    out, cur, i = [], 0, 0
    while i < len(ev):
        c, p, _ = ev[i]
        d = 0
        while i < len(ev) and ev[i][0] == c and ev[i][1] == p:
            d += ev[i][2]
            i += 1
        out.append((c, p, cur, cur + d))
        cur += d
    assert cur == 0
    return out


def bad_coverage_mask(alignment_intervals, contig_intervals, lower, upper):
    """Masked (contig, begin, end) intervals, empty ones dropped (ReferenceRegion normalises)."""
    changes = coverage_changes(alignment_intervals, contig_intervals)
    if not changes:
        return []

    def zone(c):
        return -1 if c < lower else (1 if c > upper else 0)
    acc, masking, mc, ms = [], False, 0, 0
    last = changes[0]
    for ev in changes:
        c, p, cur, new = ev
        zc, zn = zone(cur), zone(new)
        if masking and c != last[0]:
            acc.append((mc, ms, last[1]))
            masking = False
        if not masking and (zn != 0 or (zc == 0 and zc != zn)):
            masking, mc, ms = True, c, p
        elif masking and zc != 0 and zn == 0:
            acc.append((mc, ms, p))
            masking = False
        last = ev
    if masking:
        acc.append((mc, ms, last[1]))
    return [iv for iv in acc if iv[2] > iv[1]]


def max_coverage_reads(x):
    """commandline.d:1876-1884."""
    return int(x / math.log(math.log(math.log(0.1650612 * x + 5.9354533) / math.log(1.65))))


def max_improper_coverage_reads(x):
    """commandline.d:1957-1965."""
    return int(0.5 * x + math.exp(0.1875 * (8.0 - x)))


def propagate_mask(las, trace, tspace, mask_ptr, mask_iv, read_len):
    """commands/propagateMask.d:136-305 restated literally: per contig the two-pointer walk over the mask
    intervals and the alignments sorted by their begin on the contig, intervals cut to the alignment and
    translated through the trace points (floor / ceil, oracle/seq_las_trace.c restating base.d:185-203),
    mirrored for complement alignments; then the Region union per read (util/region.d:776-816).
    Returns {read: [(begin, end), ...]}."""
    import ctypes
    import numpy as np
    from . import pyoracle as oz

    def translate(la, pos, mode):
        a, b = ctypes.c_int32(), ctypes.c_int32()
        t = np.ascontiguousarray(trace[la["toff"]:la["toff"] + la["tlen"]], dtype=np.uint16)
        oz.lib().oz_translate_trace_point_a(int(la["abpos"]), int(la["aepos"]), int(la["bbpos"]), tspace, t.ctypes.data,
                                            len(t) // 2, int(pos), mode, ctypes.byref(a), ctypes.byref(b))
        return b.value
    raw = []
    contigs = sorted(set(int(a) for a in las["aread"]))
    for c in contigs:
        mi = [(int(mask_iv[j][0]), int(mask_iv[j][1])) for j in range(mask_ptr[c], mask_ptr[c + 1])]
        la_c = sorted((la for la in las if la["aread"] == c), key=lambda la: int(la["abpos"]))
        m = k = 0
        while m < len(mi) and k < len(la_c):
            la = la_c[k]
            if mi[m][1] <= la["abpos"]:
                m += 1
            elif la["aepos"] <= mi[m][0]:
                k += 1
            else:
                inter = [iv for iv in mi[m:] if iv[0] < la["aepos"]]
                for x, (b, e) in enumerate(inter):
                    if x == 0:
                        b = max(b, int(la["abpos"]))
                    if x == len(inter) - 1:
                        e = min(e, int(la["aepos"]))
                    pb, pe = translate(la, b, 0), translate(la, e, 1)
                    if la["flags"] & 1:
                        bl = int(read_len[la["bread"]])
                        pb, pe = bl - pe, bl - pb
                    raw.append((int(la["bread"]), pb, pe))
                k += 1
    raw.sort()
    out = {}
    for r, b, e in raw:
        if e <= b:
            continue
        ivs = out.setdefault(r, [])
        if ivs and b <= ivs[-1][1]:
            ivs[-1] = (ivs[-1][0], max(ivs[-1][1], e))
        else:
            ivs.append((b, e))
    return out


def validate_region(alignments, region, contig_len, region_context, window, min_coverage_reads, min_spanning_reads):
    """commands/validateRegions.d:325-512 (RegionValidator) restated literally for one region.
    alignments: (abpos, aepos) of the local alignments on the region's contig; region = (begin, end).
    Returns (numSpanningReads, weak-coverage intervals, isValid, regionWithContext)."""
    cb = region[0] - region_context if region[0] > region_context else 0      # :178-190
    ce = min(region[1] + region_context, contig_len)
    spanning = sum(1 for b, e in alignments if b < cb and ce < e)                # :409-420
    bounds = []                                                                  # :440-454
    for idx, (b, e) in enumerate(alignments):
        if b < ce and cb < e:
            bounds.append((b, 1, idx))
            bounds.append((e, -1, idx))
    bounds.sort(key=lambda t: (t[0], t[1], t[2]))
    weak, begins = [], {}
    wb, we = cb, min(cb + window, ce)                                            # window & regionWithContext
    while we <= ce and bounds:                                                   # :471-502
        trimmed = False
        for i, (pos, kind, idx) in enumerate(bounds):
            if pos < we:
                if kind == 1:
                    begins[idx] = pos
                else:
                    begins.pop(idx, None)
            else:
                bounds = bounds[i:]
                trimmed = True
                break
        n_span = sum(1 for p in begins.values() if p <= wb)
        if n_span < min_coverage_reads:
            if not weak or weak[-1][1] < wb:
                weak.append([wb, we])
            else:
                weak[-1][1] = we
        wb += 1
        we += 1
    weak = [tuple(x) for x in weak]
    return spanning, weak, spanning >= min_spanning_reads and not weak, (cb, ce)
