"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

import dentist_amd
from dentist_amd import _lib

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
    assert L.dh_abi_version() == 1


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.AlignOpts) == 64
    assert _lib.LA_DTYPE.itemsize == 48
    o = dentist_amd.default_align_opts()
    assert (o.k, o.hmin, o.band_shift, o.tspace, o.min_len, o.pen) == (14, 35, 6, 100, 500, 6)
    assert o.width <= 62


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
