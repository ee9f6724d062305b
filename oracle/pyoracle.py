"""ctypes binding of the CPU oracle (oracle/libdh_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product path under dentist_amd/.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Opts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "k", "hmin", "band_shift", "tspace", "min_len", "pen", "xdrop", "max_err_ppm", "max_cand",
        "max_la", "tcap", "strands", "skip_self", "dmax", "width", "kmer_mod", "algo")]


class La(ctypes.Structure):
    _fields_ = [("tlen", ctypes.c_int32), ("diffs", ctypes.c_int32), ("abpos", ctypes.c_int32),
                ("bbpos", ctypes.c_int32), ("aepos", ctypes.c_int32), ("bepos", ctypes.c_int32),
                ("flags", ctypes.c_uint32), ("aread", ctypes.c_int32), ("bread", ctypes.c_int32),
                ("pad", ctypes.c_int32), ("toff", ctypes.c_int64)]


LA_DTYPE = np.dtype([("tlen", "<i4"), ("diffs", "<i4"), ("abpos", "<i4"), ("bbpos", "<i4"),
                     ("aepos", "<i4"), ("bepos", "<i4"), ("flags", "<u4"), ("aread", "<i4"),
                     ("bread", "<i4"), ("pad", "<i4"), ("toff", "<i8")])


class LaSet(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("cap", ctypes.c_int64), ("la", ctypes.POINTER(La)),
                ("tn", ctypes.c_int64), ("tcap", ctypes.c_int64),
                ("trace", ctypes.POINTER(ctypes.c_uint16))]


class Db(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("off", ctypes.c_void_p), ("bases", ctypes.c_void_p),
                ("group", ctypes.c_void_p), ("mask_ptr", ctypes.c_void_p), ("mask_iv", ctypes.c_void_p),
                ("pflags", ctypes.c_void_p)]


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdh_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.oz_default_opts.argtypes = [ctypes.POINTER(Opts)]
        L.oz_align_db.argtypes = [ctypes.POINTER(Db), ctypes.POINTER(Db), ctypes.POINTER(Opts),
                                  ctypes.c_int, ctypes.POINTER(LaSet), ctypes.c_void_p]
        L.oz_align_db2.argtypes = [ctypes.POINTER(Db), ctypes.POINTER(Db), ctypes.POINTER(Opts), ctypes.c_int,
                                   ctypes.POINTER(LaSet), ctypes.POINTER(LaSet), ctypes.c_void_p]
        L.oz_la_set_init.argtypes = [ctypes.POINTER(LaSet)]
        L.oz_la_set_free.argtypes = [ctypes.POINTER(LaSet)]
        L.oz_la_set_sort.argtypes = [ctypes.POINTER(LaSet)]
        L.oz_select_best.argtypes = [ctypes.POINTER(LaSet)]
        L.oz_las_write.argtypes = [ctypes.c_char_p, ctypes.POINTER(LaSet), ctypes.c_int32]
        L.oz_las_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(LaSet), ctypes.POINTER(ctypes.c_int32)]
        L.oz_nw.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                            ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p,
                            ctypes.POINTER(ctypes.c_int32)]
        L.oz_nw.restype = ctypes.c_uint32
        L.oz_trace_points_up_to_a.restype = ctypes.c_int32
        L.oz_trace_points_up_to_a.argtypes = [ctypes.c_int32] * 5 + [ctypes.c_int]
        L.oz_trace_points_up_to_b.restype = ctypes.c_int32
        L.oz_trace_points_up_to_b.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_int]
        L.oz_translate_trace_point_a.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_int32,
                                                                        ctypes.c_int32, ctypes.c_int,
                                                                        ctypes.POINTER(ctypes.c_int32),
                                                                        ctypes.POINTER(ctypes.c_int32)]
        L.oz_local_align.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Opts),
                                     ctypes.POINTER(La), ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                     ctypes.c_void_p]
        L.oz_local_align.restype = ctypes.c_int
        _LIB = L
    return _LIB


def default_opts(**kw):
    o = Opts()
    lib().oz_default_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _db(seqdb):
    d = Db()
    d.n = seqdb.n
    d.off = seqdb.off.ctypes.data
    d.bases = seqdb.bases.ctypes.data
    d.group = seqdb.group.ctypes.data if seqdb.group is not None else None
    mask = getattr(seqdb, "mask", None)   # (ptr int64[n+1], iv int32[2m]) or None
    if mask is not None:
        d.mask_ptr = mask[0].ctypes.data
        d.mask_iv = mask[1].ctypes.data
    pf = getattr(seqdb, "pflags", None)   # uint8[n] or None: which records of a symmetric all-vs-all are wanted (oz_db.pflags)
    if pf is not None:
        d.pflags = pf.ctypes.data
    return d


def _take(ls):
    """Copy an oz_la_set into numpy (records, trace) and free it."""
    n, tn = ls.n, ls.tn
    if n:
        buf = ctypes.string_at(ls.la, n * ctypes.sizeof(La))
        las = np.frombuffer(buf, dtype=LA_DTYPE).copy()
        trace = np.ctypeslib.as_array(ls.trace, shape=(max(tn, 1),))[:tn].copy()
    else:
        las = np.zeros(0, dtype=LA_DTYPE)
        trace = np.zeros(0, dtype=np.uint16)
    lib().oz_la_set_free(ctypes.byref(ls))
    return las, trace


def align_db(A, B, opts, nthreads=1, sort=True, select_best=False):
    """Every read of B against A. Returns (records, trace u16, stats[4])."""
    L = lib()
    ls = LaSet()
    L.oz_la_set_init(ctypes.byref(ls))
    stats = np.zeros(4, dtype=np.int64)
    da, dbb = _db(A), _db(B)
    L.oz_align_db(ctypes.byref(da), ctypes.byref(dbb), ctypes.byref(opts), nthreads, ctypes.byref(ls),
                  stats.ctypes.data)
    if select_best:
        L.oz_select_best(ctypes.byref(ls))
    if sort:
        L.oz_la_set_sort(ctypes.byref(ls))
    las, trace = _take(ls)
    return las, trace, stats


def align_db_transposed(A, B, opts, nthreads=1):
    """oz_align_db2: a DH-2 mapping and the records of its transposed pairs (`damapper -C`).  Returns
    ((records, trace), (transposed records, trace)), both in LAsort order; chain flags are not set."""
    L = lib()
    ls, ls2 = LaSet(), LaSet()
    L.oz_la_set_init(ctypes.byref(ls))
    L.oz_la_set_init(ctypes.byref(ls2))
    stats = np.zeros(4, dtype=np.int64)
    da, dbb = _db(A), _db(B)
    rc = L.oz_align_db2(ctypes.byref(da), ctypes.byref(dbb), ctypes.byref(opts), nthreads, ctypes.byref(ls), ctypes.byref(ls2),
                        stats.ctypes.data)
    if rc != 0:
        raise ValueError("oz_align_db2: the transposed file is defined for algo = 1 and skip_self != 2")
    L.oz_la_set_sort(ctypes.byref(ls))
    L.oz_la_set_sort(ctypes.byref(ls2))
    return _take(ls), _take(ls2)


def nw(ref, qry, indel=1, free_shift=False):
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    ops = np.zeros(len(ref) + len(qry) + 1, dtype=np.uint8)
    nops = ctypes.c_int32(0)
    score = lib().oz_nw(ref.ctypes.data, len(ref), qry.ctypes.data, len(qry), indel, int(free_shift),
                        ops.ctypes.data, ctypes.byref(nops))
    return int(score), ops[:nops.value].copy()


def las_write(path, las, trace, tspace):
    ls = LaSet()
    arr = np.ascontiguousarray(las)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    ls.n = len(arr)
    ls.la = ctypes.cast(arr.ctypes.data, ctypes.POINTER(La))
    ls.tn = len(tr)
    ls.trace = ctypes.cast(tr.ctypes.data, ctypes.POINTER(ctypes.c_uint16))
    rc = lib().oz_las_write(path.encode(), ctypes.byref(ls), tspace)
    if rc:
        raise IOError(f"oz_las_write({path}) = {rc}")


def las_read(path):
    ls = LaSet()
    ts = ctypes.c_int32(0)
    rc = lib().oz_las_read(path.encode(), ctypes.byref(ls), ctypes.byref(ts))
    if rc:
        raise IOError(f"oz_las_read({path}) = {rc}")
    las, trace = _take(ls)
    return las, trace, ts.value


# ---------------------------------------------------------------- consensus path (consensus.c)
MAXQV = 50
MAXINS = 4
VOTE_STRIDE = 6 + 4 * MAXINS


def _la_set_view(las, trace):
    """Borrow numpy arrays as an oz_la_set (keep the arrays alive while it is used)."""
    ls = LaSet()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    ls.n = len(arr)
    ls.la = ctypes.cast(arr.ctypes.data, ctypes.POINTER(La))
    ls.tn = len(tr)
    ls.trace = ctypes.cast(tr.ctypes.data, ctypes.POINTER(ctypes.c_uint16))
    return ls, (arr, tr)


def _bind_consensus():
    L = lib()
    if getattr(L, "_cons_bound", False):
        return L
    L.oz_valid_pileup_alignment.argtypes = [ctypes.POINTER(La), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    L.oz_valid_pileup_alignment.restype = ctypes.c_int
    L.oz_tile_qv.argtypes = [ctypes.POINTER(LaSet), ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
    L.oz_rank_reference_reads.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_int32)]
    L.oz_rank_reference_reads.restype = ctypes.c_int32
    L.oz_consensus.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(Db), ctypes.POINTER(LaSet),
                               ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.oz_consensus.restype = ctypes.c_int32
    L._cons_bound = True
    return L


def tile_qv(las, trace, rlen, tspace, cov):
    L = _bind_consensus()
    rlen = np.ascontiguousarray(rlen, dtype=np.int32)
    maxtiles = int((rlen.max() + tspace - 1) // tspace) if len(rlen) else 1
    qv = np.full((len(rlen), max(maxtiles, 1)), 255, dtype=np.uint8)
    ls, keep = _la_set_view(las, trace)
    L.oz_tile_qv(ctypes.byref(ls), len(rlen), rlen.ctypes.data, tspace, cov, qv.ctypes.data, qv.shape[1])
    return qv


def rank_reference_reads(qv, rlen, tspace, bad_fraction=0.08, allowed=None):
    L = _bind_consensus()
    rlen = np.ascontiguousarray(rlen, dtype=np.int32)
    qv = np.ascontiguousarray(qv, dtype=np.uint8)
    order = np.zeros(len(rlen), dtype=np.int32)
    n = ctypes.c_int32(0)
    al = None if allowed is None else np.ascontiguousarray(allowed, dtype=np.uint8)
    bad = L.oz_rank_reference_reads(qv.ctypes.data, len(rlen), rlen.ctypes.data, tspace, qv.shape[1],
                                    al.ctypes.data if al is not None else None, bad_fraction,
                                    order.ctypes.data, ctypes.byref(n))
    return order[:n.value].copy(), int(bad)


def consensus(ref, reads, las, trace, aidx, tspace, want_votes=False):
    L = _bind_consensus()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    out = np.zeros(len(ref) * (1 + MAXINS) + 1, dtype=np.uint8)
    votes = np.zeros((len(ref), VOTE_STRIDE), dtype=np.uint32) if want_votes else None
    ls, keep = _la_set_view(las, trace)
    d = _db(reads)
    n = L.oz_consensus(ref.ctypes.data, len(ref), ctypes.byref(d), ctypes.byref(ls), aidx, tspace,
                       out.ctypes.data, votes.ctypes.data if want_votes else None)
    return (out[:n].copy(), votes) if want_votes else out[:n].copy()


def valid_pileup_alignment(la, alen, blen, allowance):
    L = _bind_consensus()
    rec = La()
    for f, _ in La._fields_:
        setattr(rec, f, int(la[f]))
    return bool(L.oz_valid_pileup_alignment(ctypes.byref(rec), alen, blen, allowance))


# ---------------------------------------------------------------- collect + process in C (pile.c)
class ProcessOpts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "ts_map", "allowance", "min_anchor", "min_reads", "max_reads", "ts_pile", "rounds", "flank_window",
        "max_align_err_ppm", "max_ins_err_ppm", "bad_fraction_ppm", "width", "dust", "algo", "max_partners",
        "min_relative_score_ppm")]


OZ_INSERTION_DTYPE = np.dtype([(n, "<i4") for n in (
    "gap", "status", "nreads", "ref_idx", "ref_read_id", "crop_left", "crop_right", "left_aepos", "right_abpos",
    "ins_begin", "ins_end", "comp", "cons_len", "left_diffs", "right_diffs", "pad")] + [("cons_off", "<i8")])


def default_process_opts(**kw):
    o = ProcessOpts()
    lib().oz_default_process_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def chain_las_c(las, min_score, min_rel_ppm=1000000):
    """oz_chain_las: chainLocalAlignments over records sorted by (aread, bread); returns the records with their chain flags
    (LAs shared by alternate chains duplicated behind their first occurrence)."""
    L = lib()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    out = ctypes.c_void_p()
    L.oz_chain_las.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    L.oz_chain_las.restype = ctypes.c_int64
    L.oz_free.argtypes = [ctypes.c_void_p]
    n = L.oz_chain_las(arr.ctypes.data, len(arr), int(min_score), int(min_rel_ppm), ctypes.byref(out))
    res = (np.frombuffer(ctypes.string_at(out, LA_DTYPE.itemsize * n), dtype=LA_DTYPE).copy() if n
           else np.zeros(0, dtype=LA_DTYPE))
    L.oz_free(out)
    return res


def collect_spanning_c(las, contigs, popts):
    """oz_collect_spanning: (gaps, [triples per gap]) -- the C twin of oracle.process.collect_spanning."""
    L = lib()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    dc = _db(contigs)
    gp, cp, tp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    npl = ctypes.c_int32(0)
    L.oz_collect_spanning.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Db), ctypes.POINTER(ProcessOpts),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32)]
    L.oz_free.argtypes = [ctypes.c_void_p]
    L.oz_collect_spanning(arr.ctypes.data, len(arr), ctypes.byref(dc), ctypes.byref(popts), ctypes.byref(gp),
                          ctypes.byref(cp), ctypes.byref(tp), ctypes.byref(npl))
    n = npl.value
    gaps = np.frombuffer(ctypes.string_at(gp, 4 * n), dtype=np.int32).copy() if n else np.zeros(0, np.int32)
    cnt = np.frombuffer(ctypes.string_at(cp, 4 * n), dtype=np.int32).copy() if n else np.zeros(0, np.int32)
    tot = int(cnt.sum())
    tri = (np.frombuffer(ctypes.string_at(tp, 12 * tot), dtype=np.int32).reshape(tot, 3).copy() if tot
           else np.zeros((0, 3), np.int32))
    for p in (gp, cp, tp):
        L.oz_free(p)
    out, at = [], 0
    for c in cnt:
        out.append(tri[at:at + c])
        at += c
    return gaps, out


def process_piles_c(contigs, reads, las, trace, gaps, triples, popts, nthreads=1):
    """oz_process_piles: every pile-up through the `process` sequence on the CPU (OpenMP over
    pile-ups).  Returns (records OZ_INSERTION_DTYPE, consensus bases)."""
    L = lib()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    g = np.ascontiguousarray(gaps, dtype=np.int32)
    cnt = np.asarray([len(t) for t in triples], dtype=np.int32)
    tri = (np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1, 3) for t in triples]))
           if len(triples) else np.zeros((0, 3), np.int32))
    out = np.zeros(len(g), dtype=OZ_INSERTION_DTYPE)
    bp = ctypes.c_void_p()
    nb = ctypes.c_int64(0)
    dc, dr = _db(contigs), _db(reads)
    L.oz_process_piles.argtypes = [ctypes.POINTER(Db), ctypes.POINTER(Db), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                   ctypes.POINTER(ProcessOpts), ctypes.c_int, ctypes.c_void_p,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    L.oz_free.argtypes = [ctypes.c_void_p]
    L.oz_process_piles(ctypes.byref(dc), ctypes.byref(dr), arr.ctypes.data, len(arr), tr.ctypes.data, g.ctypes.data,
                       cnt.ctypes.data, tri.ctypes.data, len(g), ctypes.byref(popts), nthreads, out.ctypes.data,
                       ctypes.byref(bp), ctypes.byref(nb))
    bases = np.frombuffer(ctypes.string_at(bp, nb.value), dtype=np.uint8).copy() if nb.value else np.zeros(0, np.uint8)
    L.oz_free(bp)
    return out, bases


def dust(seqdb):
    """oz_dust: low-complexity mask of every sequence as (ptr int64[n+1], iv int32 pairs)."""
    L = lib()
    d = _db(seqdb)
    ptr = np.zeros(seqdb.n + 1, dtype=np.int64)
    ivp = ctypes.c_void_p()
    L.oz_dust.argtypes = [ctypes.POINTER(Db), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    L.oz_dust.restype = ctypes.c_int64
    L.oz_free.argtypes = [ctypes.c_void_p]
    m = L.oz_dust(ctypes.byref(d), ptr.ctypes.data, ctypes.byref(ivp))
    iv = np.frombuffer(ctypes.string_at(ivp, 8 * m), dtype=np.int32).copy() if m else np.zeros(0, np.int32)
    L.oz_free(ivp)
    return ptr, iv


def with_dust(seqdb):
    """The same sequences with their dust mask attached (union with an existing mask is not needed
    by the callers: pile-up and flank DBs carry no other mask in the oracle drivers)."""
    from dentist_amd.sim import SeqDb
    out = SeqDb(seqdb.bases, seqdb.off, seqdb.group)
    ptr, iv = dust(seqdb)
    out.mask = (ptr, np.concatenate([iv, np.zeros(2, np.int32)]))
    return out


# ---------------------------------------------------------------- seed candidates of every item
CAND_DTYPE = np.dtype([("score", "<i4"), ("aseq", "<i4"), ("apos", "<i4"), ("bpos", "<i4"), ("band", "<i8")])


def seed_candidates_all(A, B, opts):
    """Candidates of every (read, strand) item of B, in item order (item = 2 * read + strand):
    (cand[nitems, max_cand] CAND_DTYPE, ncand[nitems]) -- what the seed stage hands to the extension."""
    from dentist_amd.sim import revcomp
    L = lib()
    L.oz_index_build.restype = ctypes.c_void_p
    L.oz_index_build.argtypes = [ctypes.POINTER(Db), ctypes.POINTER(Opts)]
    L.oz_index_free.argtypes = [ctypes.c_void_p]
    L.oz_seed_candidates.restype = ctypes.c_int
    L.oz_seed_candidates.argtypes = [ctypes.c_void_p, ctypes.POINTER(Db), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Opts), ctypes.c_void_p, ctypes.c_void_p]
    da = _db(A)
    ix = L.oz_index_build(ctypes.byref(da), ctypes.byref(opts))
    cand = np.zeros((2 * B.n, opts.max_cand + 1), dtype=CAND_DTYPE)
    ncand = np.zeros(2 * B.n, dtype=np.int32)
    nh = ctypes.c_int32(0)
    for r in range(B.n):
        for s in range(2):
            if not opts.strands & (1 << s):
                continue
            b = np.ascontiguousarray(B.seq(r) if s == 0 else revcomp(B.seq(r)))
            row = cand[2 * r + s]
            ncand[2 * r + s] = L.oz_seed_candidates(ix, ctypes.byref(da), b.ctypes.data, len(b),
                                                    int(B.group[r]) if B.group is not None else 0, r, 0,
                                                    ctypes.byref(opts), row.ctypes.data, ctypes.byref(nh))
    L.oz_index_free(ix)
    return cand[:, :opts.max_cand], ncand
