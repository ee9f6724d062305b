"""Seeded synthetic data (assembly, gaps, error-laden reads) for tests and bench.py.

Stands in for DAZZ_DB's ``simulator`` which the reference's tests shell out to
(tests/test-commands.sh:7-13, 102-105); workload shapes follow SURVEY.md section 8(d).
Pure host code (C++ via ctypes), no GPU involved.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdh_sim.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.dhsim_genome.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
        lib.dhsim_gaps.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        lib.dhsim_gaps.restype = ctypes.c_int32
        lib.dhsim_reads.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]
        lib.dhsim_reads.restype = ctypes.c_int64
        lib.dhsim_reads_from.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]
        lib.dhsim_reads_from.restype = ctypes.c_int64
        lib.dhsim_reads_sel.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]
        lib.dhsim_reads_sel.restype = ctypes.c_int64
        lib.dhsim_read_truth.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        _LIB = lib
    return _LIB


class SeqDb:
    """Host-side sequence collection: concatenated base codes (a,c,g,t=0..3, other=4) + offsets."""

    def __init__(self, bases, off, group=None):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        self.group = None if group is None else np.ascontiguousarray(group, dtype=np.int32)

    @property
    def n(self):
        return len(self.off) - 1

    def seq(self, i):
        return self.bases[self.off[i]:self.off[i + 1]]

    def length(self, i):
        return int(self.off[i + 1] - self.off[i])

    @staticmethod
    def from_list(seqs, group=None):
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        for i, s in enumerate(seqs):
            off[i + 1] = off[i] + len(s)
        bases = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if seqs else np.zeros(0, np.uint8)
        return SeqDb(bases, off, group)


_ENC = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("acgt"):
    _ENC[ord(_c)] = _i
    _ENC[ord(_c.upper())] = _i
_DEC = np.frombuffer(b"acgtn", dtype=np.uint8)


def encode(s):
    return _ENC[np.frombuffer(s.encode() if isinstance(s, str) else s, dtype=np.uint8)]


def decode(codes):
    return _DEC[np.minimum(np.asarray(codes, dtype=np.uint8), 4)].tobytes().decode()


def revcomp(codes):
    c = np.asarray(codes, dtype=np.uint8)[::-1]
    return np.where(c < 4, 3 - c, c).astype(np.uint8)


def genome(seed, n):
    out = np.empty(n, dtype=np.uint8)
    _lib().dhsim_genome(seed, n, out.ctypes.data)
    return out


def gaps(seed, genome_len, ngaps, minlen=50, maxlen=5000, spacing=20000):
    b = np.zeros(ngaps, dtype=np.int64)
    e = np.zeros(ngaps, dtype=np.int64)
    n = _lib().dhsim_gaps(seed, genome_len, ngaps, minlen, maxlen, spacing, b.ctypes.data, e.ctypes.data)
    return b[:n].copy(), e[:n].copy()


def reads(seed, genome_codes, nreads, mean_len, sd_len=0, min_len=100, err=0.13, p_ins=0.60, p_del=0.25, first=0,
          hp_bias=0.0, ids=None):
    """nreads reads starting with read number `first` of the stream (every read has its own RNG
    stream, so a share of the reads equals the corresponding reads of the whole set); ``ids`` = the
    reads with these numbers instead.  ``hp_bias`` > 0: homopolymer-biased indels (ONT-like profile,
    SURVEY 8(d), BASELINE configs[4])."""
    g = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        nreads, idp = len(ids), ids.ctypes.data
    off = np.zeros(nreads + 1, dtype=np.int64)
    truth = np.zeros((nreads, 3), dtype=np.int64)
    total = _lib().dhsim_reads_sel(seed, g.ctypes.data, len(g), idp, first, nreads, mean_len, sd_len, min_len, err,
                                   p_ins, p_del, hp_bias, off.ctypes.data, None, truth.ctypes.data)
    bases = np.empty(total, dtype=np.uint8)
    _lib().dhsim_reads_sel(seed, g.ctypes.data, len(g), idp, first, nreads, mean_len, sd_len, min_len, err, p_ins,
                           p_del, hp_bias, off.ctypes.data, bases.ctypes.data, truth.ctypes.data)
    return SeqDb(bases, off), truth


def read_truth(seed, genome_len, nreads, mean_len, sd_len=0, min_len=100, first=0):
    """(start, end, strand) of reads [first, first + nreads) of the stream, without their bases."""
    truth = np.zeros((nreads, 3), dtype=np.int64)
    _lib().dhsim_read_truth(seed, genome_len, first, nreads, mean_len, sd_len, min_len, truth.ctypes.data)
    return truth


def contigs_from_gaps(genome_codes, gap_begin, gap_end):
    """Split the truth sequence at the gaps: contig i = genome[end[i-1], begin[i])."""
    starts = [0] + [int(e) for e in gap_end]
    ends = [int(b) for b in gap_begin] + [len(genome_codes)]
    db = SeqDb.from_list([genome_codes[s:e] for s, e in zip(starts, ends)])
    return db, np.asarray(starts, dtype=np.int64)


class Workload:
    """A BASELINE.json-style synthetic workload (SURVEY.md 8(d) seeds: asm, +1 gaps, +2 reads)."""

    def __init__(self, genome_len, ngaps, nreads, read_len, seed=20260929, err=0.13, sd_len=0,
                 gap_min=50, gap_max=5000, spacing=20000, read_range=None, p_ins=0.60, p_del=0.25, hp_bias=0.0):
        self.truth = genome(seed, genome_len)
        self.gap_begin, self.gap_end = gaps(seed + 1, genome_len, ngaps, gap_min, gap_max, spacing)
        self.contigs, self.contig_start = contigs_from_gaps(self.truth, self.gap_begin, self.gap_end)
        # read_range = (first, end): only that share of the nreads reads is generated (sharded runs)
        self.read_first, read_end = read_range if read_range is not None else (0, nreads)
        self.nreads_total = nreads
        self.reads, self.read_truth = reads(seed + 2, self.truth, read_end - self.read_first, read_len, sd_len,
                                            err=err, first=self.read_first, p_ins=p_ins, p_del=p_del, hp_bias=hp_bias)
        self.read_profile = dict(err=err, p_ins=p_ins, p_del=p_del, hp_bias=hp_bias, sd_len=sd_len, read_len=read_len,
                                 seed=seed + 2)


class RankShare:
    """What ONE rank of ``world`` holds of a workload too large to build on one box (BASELINE configs[4]:
    3 Gb assembly, 10 000 gaps, 10 M x 20 kb reads on 8 GPUs): the whole assembly (contigs and index are
    replicated), its block of the reads (``reads``: reads [lo, hi) of the stream) and -- what
    the all-to-all of cropped reads would hand it -- the reads of ALL ranks that span the gaps it owns
    (``pile_reads``; emulation: rank r owns every world-th gap starting with gap r, and the spanning reads
    are picked by their true origin, margin 1 kb on both sides).  Reads are generated from per-read RNG
    streams, so both sets equal the corresponding reads of the full workload."""

    def __init__(self, genome_len, ngaps, nreads, read_len, rank=0, world=8, seed=20260929, err=0.10, sd_len=0,
                 p_ins=0.30, p_del=0.40, hp_bias=0.5, gap_min=50, gap_max=5000, spacing=20000, margin=1000):
        self.truth = genome(seed, genome_len)
        self.gap_begin, self.gap_end = gaps(seed + 1, genome_len, ngaps, gap_min, gap_max, spacing)
        self.contigs, self.contig_start = contigs_from_gaps(self.truth, self.gap_begin, self.gap_end)
        self.rank, self.world, self.nreads_total = rank, world, nreads
        self.read_first, read_end = nreads * rank // world, nreads * (rank + 1) // world
        prof = dict(err=err, p_ins=p_ins, p_del=p_del, hp_bias=hp_bias)
        self.reads, self.read_truth = reads(seed + 2, self.truth, read_end - self.read_first, read_len, sd_len,
                                            first=self.read_first, **prof)
        # the reads of the whole workload that span the owned gaps
        tr = read_truth(seed + 2, genome_len, nreads, read_len, sd_len)
        order = np.argsort(tr[:, 0], kind="stable")
        starts = tr[order, 0]
        maxlen = int((tr[:, 1] - tr[:, 0]).max()) if nreads else 0
        self.owned_gaps = np.arange(rank, len(self.gap_begin), world)
        ids = []
        for g in self.owned_gaps:
            lo = np.searchsorted(starts, self.gap_end[g] + margin - maxlen, side="left")
            hi = np.searchsorted(starts, self.gap_begin[g] - margin, side="right")
            cand = order[lo:hi]
            ids.append(cand[tr[cand, 1] >= self.gap_end[g] + margin])
        self.pile_read_ids = np.unique(np.concatenate(ids)) if ids else np.zeros(0, np.int64)
        self.pile_reads, self.pile_truth = reads(seed + 2, self.truth, 0, read_len, sd_len, ids=self.pile_read_ids, **prof)
