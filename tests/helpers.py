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
