"""Shared helpers for the parity tests (HIP path vs. CPU oracle)."""
import numpy as np

FIELDS = ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread")


def la_rows(las, trace):
    """Canonical python rows: tuple of record fields + the trace as a tuple."""
    rows = []
    for la in las:
        t = tuple(int(x) for x in trace[la["toff"]:la["toff"] + la["tlen"]])
        rows.append(tuple(int(la[f]) for f in FIELDS) + (t,))
    return rows


def assert_same_las(got, exp):
    """Bit-exact comparison of two (records, trace) results."""
    g, e = la_rows(*got), la_rows(*exp)
    assert len(g) == len(e), f"LA count differs: got {len(g)} expected {len(e)}"
    for i, (x, y) in enumerate(zip(g, e)):
        assert x == y, f"LA {i} differs:\n got {x}\n exp {y}"


def check_trace_invariants(las, trace, tspace):
    """base.d:434-458: sum(bbases) == bepos - bbpos, sum(diffs) == diffs, #tp from A interval."""
    for la in las:
        t = trace[la["toff"]:la["toff"] + la["tlen"]].astype(np.int64)
        assert t[1::2].sum() == la["bepos"] - la["bbpos"]
        assert t[0::2].sum() == la["diffs"]
        assert la["tlen"] // 2 == -(-int(la["aepos"]) // tspace) - int(la["abpos"]) // tspace


def plant_long_indels(w, rng):
    """Reads of the workload `w` that span a gap get a 2-5 kb insertion of foreign bases, or lose 2-5 kb of contig bases,
    1.5-3.5 kb away from the gap (every second eligible read): each then maps as two collinear records on that flank,
    ONE alignment chain (dazzler.d:1728-1758).  Returns (reads DB, indices of the changed reads)."""
    from dentist_amd import sim
    seqs = [w.reads.seq(i) for i in range(w.reads.n)]
    planted, eligible = [], 0
    for i, (s0, e0, strand) in enumerate(w.read_truth):
        for g in range(len(w.gap_begin)):
            gb, ge = int(w.gap_begin[g]), int(w.gap_end[g])
            if not (s0 + 1500 < gb and ge + 1500 < e0):
                continue
            left = gb - s0 >= e0 - ge          # the longer flank part of the read gets the indel
            if (gb - s0 if left else e0 - ge) < 7000:
                continue
            eligible += 1
            if eligible % 2:
                continue
            # a position 1.5-3.5 kb away from the gap, in read coordinates (reads are ~ (1 + ins - del) longer than the truth)
            d = int(rng.integers(1500, 3500))
            gpos = gb - d if left else ge + d
            scale = len(seqs[i]) / float(e0 - s0)
            at = int((gpos - s0) * scale) if not strand else int((e0 - gpos) * scale)
            ln = int(rng.integers(2000, 5000))
            s = seqs[i]
            if len(planted) % 2 == 0:    # foreign bases in the read
                seqs[i] = np.concatenate([s[:at], rng.integers(0, 4, ln).astype(np.uint8), s[at:]])
            else:                        # contig bases missing from the read: cut away from the gap
                lo, hi = (at - ln, at) if left != bool(strand) else (at, at + ln)
                if lo < 1000 or hi > len(s) - 1000:
                    continue
                seqs[i] = np.concatenate([s[:lo], s[hi:]])
            planted.append(i)
            break
    return sim.SeqDb.from_list(seqs), planted


def plant_gap_insertions(w, rng, ln=1500):
    """Every second read of the workload `w` that spans a gap gets `ln` foreign bases in the middle of the gap: in the
    pile-up all-vs-all it aligns with the other reads as TWO local alignments that no chain joins (indel above
    --max-indel 1000, chaining.d:434-475) -- two components of the pair.  Returns (reads DB, indices of the changed reads)."""
    from dentist_amd import sim
    seqs = [w.reads.seq(i) for i in range(w.reads.n)]
    planted, eligible = [], 0
    for i, (s0, e0, strand) in enumerate(w.read_truth):
        for g in range(len(w.gap_begin)):
            gb, ge = int(w.gap_begin[g]), int(w.gap_end[g])
            if not (s0 + 1500 < gb and ge + 1500 < e0):
                continue
            eligible += 1
            if eligible % 2:
                break
            mid = (gb + ge) // 2
            scale = len(seqs[i]) / float(e0 - s0)
            at = int((mid - s0) * scale) if not strand else int((e0 - mid) * scale)
            seqs[i] = np.concatenate([seqs[i][:at], rng.integers(0, 4, ln).astype(np.uint8), seqs[i][at:]])
            planted.append(i)
            break
    return sim.SeqDb.from_list(seqs), planted


def tandem_reads(seed=7, n=12):
    """Random reads, two of three with a tandem array planted: a unit of 24 .. 900 bases repeated so that the array spans
    at least 700 bases, every copy with 8 % errors.  Returns (SeqDb, [(read, array begin, array end, period)])."""
    from dentist_amd import sim
    rng = np.random.default_rng(seed)

    def mutate(u, err):
        out = []
        for b in u:
            x = rng.random()
            if x < err / 3:
                continue
            if x < 2 * err / 3:
                out.append(int(rng.integers(0, 4)))
            out.append(int(b) if x >= err else int(rng.integers(0, 4)))
        return np.array(out, dtype=np.uint8)
    seqs, truth = [], []
    periods = [24, 150, 400, 900, 60]
    for i in range(n):
        ln = int(rng.integers(6000, 12000))
        s = rng.integers(0, 4, ln).astype(np.uint8)
        if i % 3 != 2:
            per = periods[len(truth) % len(periods)]
            copies = max(int(rng.integers(4, 12)), 700 // per + 2)
            unit = rng.integers(0, 4, per).astype(np.uint8)
            arr = np.concatenate([mutate(unit, 0.08) for _ in range(copies)])
            at = int(rng.integers(1000, ln - 1000))
            s = np.concatenate([s[:at], arr, s[at:]])
            truth.append((i, at, at + len(arr), per))
        seqs.append(s)
    return sim.SeqDb.from_list(seqs), truth
