"""ctypes binding of include/dentist_hip.h (libdentist_hip.so). No compute, no fallback."""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # DH_DEV_LIB: development builds of the same library (profiling counters compiled in), never a fallback
    return os.environ.get("DH_DEV_LIB") or os.path.join(_HERE, "libdentist_hip.so")


class DhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdentist_hip error {code}: {msg}")
        self.code = code


class AlignOpts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "k", "hmin", "band_shift", "tspace", "min_len", "pen", "xdrop", "max_err_ppm", "max_cand",
        "max_la", "tcap", "strands", "skip_self", "dmax", "width", "kmer_mod", "algo")]


class AlignStats(ctypes.Structure):
    _fields_ = [("hits", ctypes.c_int64), ("cands", ctypes.c_int64), ("alignments", ctypes.c_int64),
                ("wave_cells", ctypes.c_int64), ("las", ctypes.c_int64), ("b_bases", ctypes.c_int64),
                ("ms_index", ctypes.c_float), ("ms_seed", ctypes.c_float), ("ms_wave", ctypes.c_float),
                ("ms_gather", ctypes.c_float), ("ms_total", ctypes.c_float),
                ("wave_launches", ctypes.c_int32), ("overflow_items", ctypes.c_int32), ("big_items", ctypes.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CumStats(ctypes.Structure):
    _fields_ = [("ms_index", ctypes.c_double), ("ms_seed", ctypes.c_double), ("ms_wave", ctypes.c_double),
                ("ms_gather", ctypes.c_double)] + [(n, ctypes.c_int64) for n in (
                    "wave_launches", "wave_cells", "alignments", "las", "aligned_bp", "trace_values", "hits",
                    "b_bases")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ProcessOpts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "tspace_map", "allowance", "min_anchor", "min_reads", "max_reads", "tspace_pile", "rounds",
        "flank_window", "max_align_err_ppm", "max_ins_err_ppm", "bad_fraction_ppm", "width", "dust", "algo",
        "max_partners", "min_relative_score_ppm")]


INSERTION_DTYPE = np.dtype([(n, "<i4") for n in (
    "contig_left", "status", "nreads", "ref_read", "ref_read_id", "crop_left", "crop_right", "left_aepos",
    "right_abpos", "ins_begin", "ins_end", "comp", "cons_len", "left_diffs", "right_diffs", "join")]
    + [("cons_off", "<i8"), ("contig_right", "<i4"), ("pad", "<i4")])

LA_DTYPE = np.dtype([("tlen", "<i4"), ("diffs", "<i4"), ("abpos", "<i4"), ("bbpos", "<i4"),
                     ("aepos", "<i4"), ("bepos", "<i4"), ("flags", "<u4"), ("aread", "<i4"),
                     ("bread", "<i4"), ("pad", "<i4"), ("toff", "<i8")])

# every symbol include/dentist_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "dh_last_error", "dh_abi_version", "dh_ctx_create", "dh_ctx_destroy", "dh_ctx_sync",
    "dh_default_align_opts", "dh_db_create", "dh_db_destroy", "dh_db_drop_cache", "dh_db_nreads",
    "dh_db_total_bases", "dh_la_set_destroy", "dh_la_set_count", "dh_la_set_trace_len",
    "dh_la_set_records", "dh_la_set_trace", "dh_la_set_tspace", "dh_get_align_stats", "dh_get_cum_stats", "dh_get_mjoin_counts", "dh_ctx_release_scratch",
    "dh_align_db",
    "dh_las_write", "dh_las_read", "dh_default_process_opts", "dh_collect_spanning", "dh_pileups_destroy",
    "dh_pileups_count", "dh_pileups_get", "dh_process_pileups", "dh_insertions_destroy",
    "dh_insertions_count", "dh_insertions_records", "dh_insertions_bases", "dh_insertions_bases_len",
    "dh_get_process_stats", "dh_dazz_create_dam", "dh_dazz_create_db", "dh_dazz_split", "dh_dazz_open",
    "dh_dazz_close", "dh_dazz_nreads", "dh_dazz_first_id", "dh_dazz_bases", "dh_dazz_offsets",
    "dh_dazz_origin", "dh_dazz_fpulse", "dh_dazz_header", "dh_dazz_read_mask", "dh_dazz_write_mask",
    "dh_db_set_mask", "dh_output_fasta", "dh_tile_qv", "dh_consensus", "dh_las_merge",
    "dh_collect_candidates", "dh_pileups_create", "dh_pileups_select", "dh_align_db_block", "dh_la_set_merge",
    "dh_crop_pileups", "dh_cropped_create", "dh_cropped_destroy", "dh_cropped_npiles", "dh_cropped_records",
    "dh_cropped_nreads", "dh_cropped_pile", "dh_cropped_entry", "dh_cropped_read_id", "dh_cropped_offsets",
    "dh_cropped_bases", "dh_process_cropped", "dh_translate_trace_point", "dh_db_dust", "dh_db_get_mask",
    "dh_dazz_write_track", "dh_dazz_read_track", "dh_dazz_remove", "dh_dazz_flags",
    "dh_pileupdb_write", "dh_pileupdb_read", "dh_insertiondb_write", "dh_insertiondb_read", "dh_insertiondb_merge", "dh_chaindb_destroy",
    "dh_chaindb_npiles", "dh_chaindb_pile_counts", "dh_chaindb_nread_alignments", "dh_chaindb_read_alignment_counts",
    "dh_chaindb_nseeded", "dh_chaindb_seeded", "dh_chaindb_nlas", "dh_chaindb_las", "dh_chaindb_ntrace",
    "dh_chaindb_trace", "dh_chaindb_ninsertions", "dh_chaindb_insertions", "dh_chaindb_bases", "dh_chaindb_read_ids",
    "dh_insertions_write_db", "dh_pileups_write_db", "dh_pileups_flat", "dh_collect_filter",
    "dh_map_reads", "dh_validate_regions", "dh_propagate_mask", "dh_db_mask_coverage", "dh_max_coverage_reads", "dh_max_improper_coverage_reads",
    "dh_default_scaffold_opts", "dh_scaffold_pileups", "dh_scaffold_npiles", "dh_scaffold_nentries", "dh_scaffold_joins",
    "dh_scaffold_entries", "dh_scaffold_destroy", "dh_scaffold_spanning", "dh_scaffold_gap_pileups", "dh_cropped_create2",
    "dh_cropped_kind", "dh_get_process_work", "dh_scaffold_pileups_cb", "dh_scaffold_pileups_resolved", "dh_comm_unique_id", "dh_comm_create", "dh_comm_create_local", "dh_comm_destroy",
    "dh_comm_rank", "dh_comm_world", "dh_comm_all_gather", "dh_comm_all_to_all", "dh_shard_run", "dh_set_near_best", "dh_ctx_set_near_best", "dh_shard_free", "dh_shard_pack_candidates", "dh_shard_plan_create", "dh_shard_read_joins", "dh_align_db_transposed", "dh_remap_skipping_reads", "dh_shard_graph_plan_create", "dh_shard_plan_destroy",
    "dh_shard_plan_las", "dh_shard_plan_nlas", "dh_shard_plan_pileups", "dh_shard_plan_owner", "dh_shard_pack_cropped",
    "dh_shard_unpack_cropped", "dh_insertions_read_ids", "dh_insertions_read_ids_off", "dh_output_assembly", "dh_default_output_opts",
    "dh_common_trace_point", "dh_pileups_create_joins", "dh_pileups_get_join", "dh_scaffold_all_pileups",
    "dh_crop_pileups_masked", "dh_process_pileups_masked", "dh_process_pileups_set", "dh_la_set_trace_on_device", "dh_scaffold_graph_probe",
]

_LIB = None


def lib():
    """Load libdentist_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `make` or __graft_entry__.build(); "
                           "dentist_amd has no CPU fallback")
    L = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.dh_last_error.restype = ctypes.c_char_p
    L.dh_abi_version.restype = i32
    L.dh_ctx_create.argtypes = [i32, vp, ctypes.POINTER(vp)]
    L.dh_ctx_destroy.argtypes = [vp]
    L.dh_ctx_sync.argtypes = [vp]
    L.dh_default_align_opts.argtypes = [ctypes.POINTER(AlignOpts)]
    L.dh_db_create.argtypes = [vp, vp, vp, i32, vp, ctypes.POINTER(vp)]
    L.dh_db_destroy.argtypes = [vp]
    L.dh_db_drop_cache.argtypes = [vp]
    L.dh_db_nreads.argtypes = [vp]
    L.dh_db_nreads.restype = i32
    L.dh_db_total_bases.argtypes = [vp]
    L.dh_db_total_bases.restype = i64
    L.dh_la_set_destroy.argtypes = [vp]
    L.dh_la_set_count.argtypes = [vp]
    L.dh_la_set_count.restype = i64
    L.dh_la_set_trace_len.argtypes = [vp]
    L.dh_la_set_trace_len.restype = i64
    L.dh_la_set_records.argtypes = [vp]
    L.dh_la_set_records.restype = vp
    L.dh_la_set_trace.argtypes = [vp]
    L.dh_la_set_trace.restype = vp
    L.dh_la_set_tspace.argtypes = [vp]
    L.dh_la_set_tspace.restype = i32
    L.dh_get_align_stats.argtypes = [vp, ctypes.POINTER(AlignStats)]
    L.dh_get_cum_stats.argtypes = [vp, ctypes.POINTER(CumStats), i32]
    L.dh_align_db.argtypes = [vp, vp, vp, ctypes.POINTER(AlignOpts), i32, ctypes.POINTER(vp)]
    L.dh_align_db_block.argtypes = [vp, vp, vp, i32, i32, ctypes.POINTER(AlignOpts), i32, ctypes.POINTER(vp)]
    L.dh_la_set_merge.argtypes = [ctypes.POINTER(vp), i32, ctypes.POINTER(vp)]
    L.dh_las_write.argtypes = [ctypes.c_char_p, vp, i64, vp, i32]
    L.dh_las_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.dh_default_process_opts.argtypes = [ctypes.POINTER(ProcessOpts)]
    L.dh_collect_spanning.argtypes = [vp, i64, vp, i32, ctypes.POINTER(ProcessOpts), ctypes.POINTER(vp)]
    L.dh_collect_candidates.argtypes = L.dh_collect_spanning.argtypes
    L.dh_pileups_create.argtypes = [vp, vp, i32, vp, ctypes.POINTER(vp)]
    L.dh_pileups_select.argtypes = [vp, vp, i64, ctypes.POINTER(ProcessOpts), ctypes.POINTER(vp)]
    L.dh_pileups_destroy.argtypes = [vp]
    L.dh_pileups_count.argtypes = [vp]
    L.dh_pileups_count.restype = i32
    L.dh_pileups_get.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(vp)]
    L.dh_pileups_get.restype = i32
    L.dh_process_pileups.argtypes = [vp, vp, vp, vp, i64, vp, vp, ctypes.POINTER(ProcessOpts), ctypes.POINTER(vp)]
    L.dh_crop_pileups.argtypes = [vp, vp, vp, i32, vp, i64, vp, vp, ctypes.POINTER(ProcessOpts), ctypes.POINTER(vp)]
    L.dh_cropped_create.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.dh_cropped_create2.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.dh_cropped_kind.argtypes = [vp]
    L.dh_cropped_kind.restype = vp
    L.dh_cropped_destroy.argtypes = [vp]
    for fn in (L.dh_cropped_npiles, L.dh_cropped_nreads):
        fn.argtypes = [vp]
        fn.restype = i32
    for fn in (L.dh_cropped_records, L.dh_cropped_pile, L.dh_cropped_entry, L.dh_cropped_read_id,
               L.dh_cropped_offsets, L.dh_cropped_bases):
        fn.argtypes = [vp]
        fn.restype = vp
    L.dh_process_cropped.argtypes = [vp, vp, vp, ctypes.POINTER(ProcessOpts), ctypes.POINTER(vp)]
    L.dh_translate_trace_point.argtypes = [vp, vp, i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.dh_insertions_destroy.argtypes = [vp]
    L.dh_insertions_count.argtypes = [vp]
    L.dh_insertions_count.restype = i32
    L.dh_insertions_records.argtypes = [vp]
    L.dh_insertions_records.restype = vp
    L.dh_insertions_bases.argtypes = [vp]
    L.dh_insertions_bases.restype = vp
    L.dh_insertions_bases_len.argtypes = [vp]
    L.dh_insertions_bases_len.restype = i64
    L.dh_get_process_stats.argtypes = [vp, vp, vp]
    L.dh_dazz_create_dam.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64]
    L.dh_dazz_create_db.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64]
    L.dh_dazz_split.argtypes = [ctypes.c_char_p, i32, i32, i64]
    L.dh_dazz_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.dh_dazz_close.argtypes = [vp]
    L.dh_dazz_nreads.argtypes = [vp]
    L.dh_dazz_nreads.restype = i32
    L.dh_dazz_first_id.argtypes = [vp]
    L.dh_dazz_first_id.restype = i32
    for fn in (L.dh_dazz_bases, L.dh_dazz_offsets, L.dh_dazz_origin, L.dh_dazz_fpulse):
        fn.argtypes = [vp]
        fn.restype = vp
    L.dh_dazz_header.argtypes = [vp, i32]
    L.dh_dazz_read_mask.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, vp, vp, i64]
    L.dh_dazz_read_mask.restype = i64
    L.dh_dazz_write_mask.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i32, vp, vp]
    L.dh_db_set_mask.argtypes = [vp, vp, vp]
    L.dh_db_dust.argtypes = [vp]
    L.dh_db_get_mask.argtypes = [vp, vp, vp, i64]
    L.dh_db_get_mask.restype = i64
    L.dh_las_merge.argtypes = [ctypes.POINTER(ctypes.c_char_p), i32, ctypes.c_char_p]
    L.dh_tile_qv.argtypes = [vp, vp, vp, i64, vp, i32, i32, vp, i32]
    L.dh_consensus.argtypes = [vp, vp, vp, i64, vp, i32, i32, i32, vp, i64, ctypes.POINTER(i64)]
    L.dh_output_fasta.argtypes = [ctypes.c_char_p, ctypes.c_char_p, vp, vp, i32, vp, ctypes.POINTER(ctypes.c_char_p),
                                  vp, vp, i32, vp, i32, i32]
    L.dh_dazz_header.restype = ctypes.c_char_p
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise DhError(rc, lib().dh_last_error().decode(errors="replace"))


def default_align_opts(**kw):
    o = AlignOpts()
    lib().dh_default_align_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class _LaSetOwner:
    """Keeps a library-owned dh_la_set alive for as long as numpy views of its buffers exist."""

    def __init__(self, h):
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().dh_la_set_destroy(self._h)
                self._h = None
        except Exception:
            pass


def _take_la_set(h):
    """Zero-copy numpy views of the (page-locked) buffers of a dh_la_set; the set is destroyed
    when the last view goes away."""
    L = lib()
    n, tn = L.dh_la_set_count(h), L.dh_la_set_trace_len(h)
    ts = L.dh_la_set_tspace(h)
    owner = _LaSetOwner(h)
    if n:
        buf = (ctypes.c_uint8 * (n * LA_DTYPE.itemsize)).from_address(L.dh_la_set_records(h))
        buf._owner = owner
        las = np.frombuffer(buf, dtype=LA_DTYPE)
    else:
        las = np.zeros(0, dtype=LA_DTYPE)
    if tn:
        buf = (ctypes.c_uint16 * tn).from_address(L.dh_la_set_trace(h))
        buf._owner = owner
        trace = np.frombuffer(buf, dtype=np.uint16)
    else:
        trace = np.zeros(0, dtype=np.uint16)
    return las, trace, ts


class Context:
    """One per process / GPU. ``stream``: a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        h = ctypes.c_void_p()
        _check(lib().dh_ctx_create(device, ctypes.c_void_p(stream) if stream else None, ctypes.byref(h)))
        self._h = h
        self._dbs = []

    def close(self):
        if self._h:
            for ref in self._dbs:  # device DBs must go before their context
                d = ref()
                if d is not None:
                    d.close()
            self._dbs = []
            lib().dh_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().dh_ctx_sync(self._h))

    def db(self, seqdb):
        return Db(self, seqdb)

    def align_stats(self):
        st = AlignStats()
        _check(lib().dh_get_align_stats(self._h, ctypes.byref(st)))
        return st

    def release_scratch(self):
        """dh_ctx_release_scratch: hand the context's grow-only device scratch back (the next call allocates again)."""
        lib().dh_ctx_release_scratch.argtypes = [ctypes.c_void_p]
        _check(lib().dh_ctx_release_scratch(self._h))

    def mjoin_counts(self, reset=False):
        """(chunks seeded by the partitioned k-mer join, chunks redone by the directory lookups) of this context's mapping calls."""
        out = (ctypes.c_int64 * 2)()
        lib().dh_get_mjoin_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        _check(lib().dh_get_mjoin_counts(self._h, out, int(reset)))
        return int(out[0]), int(out[1])

    def cum_stats(self, reset=False):
        st = CumStats()
        _check(lib().dh_get_cum_stats(self._h, ctypes.byref(st), int(reset)))
        return st

    def remap_skipping_reads(self, contigs, reads, contig_ids, read_ids, opts, allowance):
        """dh_remap_skipping_reads: the re-mapping call of `resolveBubbles` (pileups.d:1316-1385)."""
        ci = np.ascontiguousarray(contig_ids, dtype=np.int32)
        ri = np.ascontiguousarray(read_ids, dtype=np.int32)
        h = ctypes.c_void_p()
        L = lib()
        L.dh_remap_skipping_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                              ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                              ctypes.POINTER(ctypes.c_void_p)]
        _check(L.dh_remap_skipping_reads(self._h, contigs._h, reads._h, ci.ctypes.data, len(ci), ri.ctypes.data, len(ri),
                                         ctypes.byref(opts), int(allowance), ctypes.byref(h)))
        las, trace, _ = _take_la_set(h)
        return las, trace

    def align_db_transposed(self, A, B, opts, select_best=False):
        """dh_align_db_transposed (`damapper -C`): ((records, trace), (transposed records, trace)), LAsort order."""
        h, h2 = ctypes.c_void_p(), ctypes.c_void_p()
        _check(lib().dh_align_db_transposed(self._h, A._h, B._h, ctypes.byref(opts), int(select_best), ctypes.byref(h),
                                            ctypes.byref(h2)))
        las, trace, _ = _take_la_set(h)
        las2, trace2, _ = _take_la_set(h2)
        return (las, trace), (las2, trace2)

    def align_db(self, A, B, opts, select_best=False):
        """Every read of B against A on the GPU. Returns (records, trace u16) in LAsort order."""
        h = ctypes.c_void_p()
        _check(lib().dh_align_db(self._h, A._h, B._h, ctypes.byref(opts), int(select_best), ctypes.byref(h)))
        las, trace, _ = _take_la_set(h)
        return las, trace


    def map_reads(self, A, B, opts, popts, first=0, count=None, repeat_mask=None, sorted=True, candidates=False,
                  trace_on_device=False):
        """dh_map_reads: the mapping pass (chain flags as with select_best) with the six `collect` filters
        applied chunk by chunk on the host while the device maps on.  Returns (las, trace, dropped[6]);
        with sorted=False, candidates=True also the spanning-read candidates (Pileups, not yet cut).
        trace_on_device=True: `trace` is a DeviceTrace -- the values stay in HBM (process_pileups takes it as it is and
        fetches what the cropper reads; .numpy() downloads everything)."""
        return _map_reads(self, A, B, opts, popts, first, count, repeat_mask, sorted, candidates, trace_on_device)

    def align_db_block(self, A, B, first, count, opts, select_best=False, raw=False):
        """`damapper ref reads.<block>`: reads [first, first + count) of B against A.  raw=True
        returns the library handle (for merge_las) instead of numpy views."""
        h = ctypes.c_void_p()
        _check(lib().dh_align_db_block(self._h, A._h, B._h, first, count, ctypes.byref(opts), int(select_best),
                                       ctypes.byref(h)))
        if raw:
            return h
        las, trace, _ = _take_la_set(h)
        return las, trace


class DeviceTrace:
    """The trace values of a mapping result left on the device (dh_map_reads, want_sorted & 8).  Keeps the result set
    alive; numpy() is the host copy (downloaded on first use)."""

    def __init__(self, owner, handle):
        self._owner, self._h = owner, handle

    def numpy(self):
        L = lib()
        tn = L.dh_la_set_trace_len(self._h)
        if not tn:
            return np.zeros(0, dtype=np.uint16)
        buf = (ctypes.c_uint16 * tn).from_address(L.dh_la_set_trace(self._h))
        buf._owner = self._owner
        return np.frombuffer(buf, dtype=np.uint16)

    def on_device(self):
        L = lib()
        L.dh_la_set_trace_on_device.argtypes = [ctypes.c_void_p]
        return bool(L.dh_la_set_trace_on_device(self._h))

    def __len__(self):
        return int(lib().dh_la_set_trace_len(self._h))


def _map_reads(ctx, A, B, opts, popts, first=0, count=None, repeat_mask=None, sorted=True, candidates=False,
               trace_on_device=False):
    rp = ri = None
    if repeat_mask is not None:
        rp = np.ascontiguousarray(repeat_mask[0], dtype=np.int64)
        ri = np.ascontiguousarray(np.concatenate([repeat_mask[1], [0, 0]]), dtype=np.int32)
    L = lib()
    n = L.dh_db_nreads(B._h)
    count = n - first if count is None else count
    dropped = np.zeros(6, dtype=np.int64)
    h, ph = ctypes.c_void_p(), ctypes.c_void_p()
    L.dh_map_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                               ctypes.POINTER(AlignOpts), ctypes.POINTER(ProcessOpts), ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_void_p)]
    _check(L.dh_map_reads(ctx._h, A._h, B._h, int(first), int(count), ctypes.byref(opts), ctypes.byref(popts),
                          rp.ctypes.data if rp is not None else None, ri.ctypes.data if ri is not None else None,
                          (1 if sorted else 0) | (8 if trace_on_device else 0), dropped.ctypes.data, ctypes.byref(h),
                          ctypes.byref(ph) if candidates else None))
    if trace_on_device:
        n = L.dh_la_set_count(h)
        owner = _LaSetOwner(h)
        if n:
            buf = (ctypes.c_uint8 * (n * LA_DTYPE.itemsize)).from_address(L.dh_la_set_records(h))
            buf._owner = owner
            las = np.frombuffer(buf, dtype=LA_DTYPE)
        else:
            las = np.zeros(0, dtype=LA_DTYPE)
        trace = DeviceTrace(owner, h)
    else:
        las, trace, _ = _take_la_set(h)
    if candidates:
        return las, trace, dropped, Pileups(None, None, None, _handle=ph)
    return las, trace, dropped


def merge_las(handles):
    """LAmerge of in-memory block results (handles from align_db_block(raw=True), consumed)."""
    arr = (ctypes.c_void_p * len(handles))(*[h.value for h in handles])
    out = ctypes.c_void_p()
    rc = lib().dh_la_set_merge(arr, len(handles), ctypes.byref(out))
    for h in handles:
        lib().dh_la_set_destroy(h)
    _check(rc)
    las, trace, _ = _take_la_set(out)
    return las, trace


class Db:
    """Device-resident sequence DB (HBM); built from a dentist_amd.sim.SeqDb-like object."""

    def __init__(self, ctx, seqdb):
        self.ctx = ctx
        bases = np.ascontiguousarray(seqdb.bases, dtype=np.uint8)
        off = np.ascontiguousarray(seqdb.off, dtype=np.int64)
        mask = getattr(seqdb, "mask", None)
        group = None if getattr(seqdb, "group", None) is None else np.ascontiguousarray(seqdb.group, np.int32)
        h = ctypes.c_void_p()
        _check(lib().dh_db_create(ctx._h, bases.ctypes.data, off.ctypes.data, len(off) - 1,
                                  group.ctypes.data if group is not None else None, ctypes.byref(h)))
        self._h = h
        ctx._dbs.append(weakref.ref(self))
        if mask is not None:
            self.set_mask(mask[0], mask[1])

    def drop_cache(self):
        _check(lib().dh_db_drop_cache(self._h))

    def set_mask(self, ptr, iv):
        """Soft mask (union of -m tracks): ptr int64[n+1], iv int32 (begin, end) pairs.  Replaces the tracks
        of an earlier call; the bits of dust() / mask_coverage() are a layer of their own and stay (the
        effective mask is the OR).  None clears everything."""
        if ptr is None:
            _check(lib().dh_db_set_mask(self._h, None, None))
            return
        p = np.ascontiguousarray(ptr, dtype=np.int64)
        v = np.ascontiguousarray(iv, dtype=np.int32)
        _check(lib().dh_db_set_mask(self._h, p.ctypes.data, v.ctypes.data))

    def dust(self):
        """DBdust on the device: low-complexity windows are ORed into the soft mask."""
        _check(lib().dh_db_dust(self._h))

    def mask_coverage(self, las, lower, upper, read_off=None, improper_only=False, allowance=0):
        """dh_db_mask_coverage: regions with alignment coverage outside [lower, upper] join the soft mask
        (maskRepetitiveRegions.d:238-430); improper_only counts improper alignments only."""
        arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
        ro = np.ascontiguousarray(read_off, dtype=np.int64) if read_off is not None else None
        L = lib()
        L.dh_db_mask_coverage.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        _check(L.dh_db_mask_coverage(self._h, arr.ctypes.data if len(arr) else None, len(arr),
                                     ro.ctypes.data if ro is not None else None, len(ro) - 1 if ro is not None else 0,
                                     int(lower), int(upper), 1 if improper_only else 0, int(allowance)))

    def get_mask(self):
        """The soft mask as (ptr int64[n+1], iv int32 (begin, end) pairs)."""
        n = lib().dh_db_nreads(self._h)
        ptr = np.zeros(n + 1, dtype=np.int64)
        m = lib().dh_db_get_mask(self._h, ptr.ctypes.data, None, 0)
        if m < 0:
            _check(int(m))
        iv = np.zeros(max(2 * m, 2), dtype=np.int32)
        lib().dh_db_get_mask(self._h, ptr.ctypes.data, iv.ctypes.data, m)
        return ptr, iv[:2 * m]

    def close(self):
        if self._h:
            if self.ctx._h:
                lib().dh_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def translate_trace_point(la, trace, tspace, apos, mode="floor"):
    """Trace.translateTracePoint!"contigA" (base.d:185-244) of the product: (a, b) of the trace point
    apos is assigned to; raises DhError when apos is outside the LA.  la: one LA_DTYPE record."""
    rec = np.zeros(1, dtype=LA_DTYPE)
    rec[0] = la
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    a, b = ctypes.c_int32(), ctypes.c_int32()
    _check(lib().dh_translate_trace_point(rec.ctypes.data, tr.ctypes.data, tspace, apos,
                                          {"floor": 0, "ceil": 1}[mode], ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def common_trace_point(las, first, contig_len, tspace, seed_front, mask=None):
    """getCommonTracePoint (cropper.d:446-500) of the product for one flank: first = the first records of the flank's
    alignment chains in las; mask = [(begin, end), ...] of the contig's repeat mask.  -1 when there is none."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    fi = np.ascontiguousarray(first, dtype=np.int32)
    mk = np.ascontiguousarray(mask if mask is not None else [], dtype=np.int32).reshape(-1)
    out = ctypes.c_int32()
    L = lib()
    L.dh_common_trace_point.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.POINTER(ctypes.c_int32)]
    _check(L.dh_common_trace_point(arr.ctypes.data, len(arr), fi.ctypes.data, len(fi), contig_len, tspace,
                                   int(bool(seed_front)), mk.ctypes.data if len(mk) else None, len(mk) // 2, ctypes.byref(out)))
    return out.value


def las_write(path, las, trace, tspace):
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    _check(lib().dh_las_write(path.encode(), arr.ctypes.data, len(arr), tr.ctypes.data, tspace))


def las_read(path):
    h = ctypes.c_void_p()
    _check(lib().dh_las_read(path.encode(), ctypes.byref(h)))
    return _take_la_set(h)


def default_process_opts(**kw):
    o = ProcessOpts()
    lib().dh_default_process_opts(ctypes.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Pileups:
    """Spanning-read pile-ups per gap (host only): dh_collect_spanning; ``candidates=True`` skips the
    min/max-reads cut (dh_collect_candidates)."""

    def __init__(self, las, contig_off, opts, candidates=False, _handle=None, _keep=None):
        self._keep = _keep   # a borrowed handle (owned by _keep, e.g. a ShardPlan): never destroyed here
        if _handle is not None:
            self._h = _handle
            return
        arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
        off = np.ascontiguousarray(contig_off, dtype=np.int64)
        h = ctypes.c_void_p()
        fn = lib().dh_collect_candidates if candidates else lib().dh_collect_spanning
        _check(fn(arr.ctypes.data, len(arr), off.ctypes.data, len(off) - 1, ctypes.byref(opts), ctypes.byref(h)))
        self._h = h

    @classmethod
    def from_triples(cls, contig_left, triples_per_pile):
        """dh_pileups_create: gaps ordered by contig_left, one (k, 3) int32 array of
        (read, left LA index, right LA index) per gap."""
        cl = np.ascontiguousarray(contig_left, dtype=np.int32)
        cnt = np.asarray([len(t) for t in triples_per_pile], dtype=np.int32)
        tri = (np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1, 3)
                                                    for t in triples_per_pile]), dtype=np.int32)
               if len(triples_per_pile) else np.zeros((0, 3), np.int32))
        h = ctypes.c_void_p()
        _check(lib().dh_pileups_create(cl.ctypes.data, cnt.ctypes.data, len(cl), tri.ctypes.data, ctypes.byref(h)))
        return cls(None, None, None, _handle=h)

    @classmethod
    def from_joins(cls, nodes4, triples_per_pile):
        """dh_pileups_create_joins: pile-ups of any join -- nodes4[i] = (contig0, seed0, contig1, seed1), seed 0 = front,
        1 = back, contig1 = -1 for an extension pile-up; ordered by their nodes."""
        nd = np.ascontiguousarray(nodes4, dtype=np.int32).reshape(-1, 4)
        cnt = np.asarray([len(t) for t in triples_per_pile], dtype=np.int32)
        tri = (np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1, 3)
                                                    for t in triples_per_pile]), dtype=np.int32)
               if len(triples_per_pile) else np.zeros((0, 3), np.int32))
        h = ctypes.c_void_p()
        L = lib()
        L.dh_pileups_create_joins.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        _check(L.dh_pileups_create_joins(nd.ctypes.data, cnt.ctypes.data, len(nd), tri.ctypes.data, ctypes.byref(h)))
        return cls(None, None, None, _handle=h)

    def get_join(self, i):
        """(contig0, seed0, contig1, seed1) of pile-up i."""
        nd = np.zeros(4, dtype=np.int32)
        L = lib()
        L.dh_pileups_get_join.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        _check(L.dh_pileups_get_join(self._h, i, nd.ctypes.data))
        return tuple(int(x) for x in nd)

    def write_db(self, path, las, trace, contig_off, read_off, tspace=100):
        """dh_pileups_write_db: DENTIST's pile-ups.db for these pile-ups."""
        arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
        tr = np.ascontiguousarray(trace, dtype=np.uint16)
        co, ro = np.ascontiguousarray(contig_off, dtype=np.int64), np.ascontiguousarray(read_off, dtype=np.int64)
        L = lib()
        L.dh_pileups_write_db.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p]
        _check(L.dh_pileups_write_db(self._h, arr.ctypes.data, len(arr), tr.ctypes.data, co.ctypes.data, len(co) - 1,
                                     ro.ctypes.data, len(ro) - 1, tspace, path.encode()))

    def flat(self):
        """(contig_left int32[npiles], count int32[npiles], triples int32[total, 3]) in one call."""
        L = lib()
        L.dh_pileups_flat.argtypes = [ctypes.c_void_p] * 4
        L.dh_pileups_flat.restype = ctypes.c_int64
        n = len(self)
        cl, cnt = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        tot = L.dh_pileups_flat(self._h, None, None, None)
        tri = np.zeros((tot, 3), dtype=np.int32)
        L.dh_pileups_flat(self._h, cl.ctypes.data, cnt.ctypes.data, tri.ctypes.data)
        return cl, cnt, tri

    @classmethod
    def from_flat(cls, contig_left, count, triples):
        cl = np.ascontiguousarray(contig_left, dtype=np.int32)
        cnt = np.ascontiguousarray(count, dtype=np.int32)
        tri = np.ascontiguousarray(triples, dtype=np.int32)
        h = ctypes.c_void_p()
        _check(lib().dh_pileups_create(cl.ctypes.data, cnt.ctypes.data, len(cl), tri.ctypes.data, ctypes.byref(h)))
        return cls(None, None, None, _handle=h)

    def select(self, las, opts):
        """dh_pileups_select: the min_reads / max_reads cut."""
        arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
        h = ctypes.c_void_p()
        _check(lib().dh_pileups_select(self._h, arr.ctypes.data, len(arr), ctypes.byref(opts), ctypes.byref(h)))
        return Pileups(None, None, None, _handle=h)

    def __len__(self):
        return lib().dh_pileups_count(self._h)

    def get(self, i):
        g = ctypes.c_int32()
        p = ctypes.c_void_p()
        n = lib().dh_pileups_get(self._h, i, ctypes.byref(g), ctypes.byref(p))
        tri = np.frombuffer(ctypes.string_at(p, n * 12), dtype=np.int32).reshape(n, 3).copy()
        return g.value, tri

    def close(self):
        if self._h:
            if getattr(self, "_keep", None) is None:
                lib().dh_pileups_destroy(self._h)
            self._h = None
            self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- sharded path: the host work between the collectives
_KEEP = {}


def _blob(ptr, n):
    """numpy copy of a malloc'd blob of the library."""
    if not n:
        return np.zeros(0, np.uint8)
    addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
    return np.frombuffer((ctypes.c_uint8 * n).from_address(addr), dtype=np.uint8).copy()   # one copy


def shard_pack_candidates(cands, las, read_shift=0):
    """dh_shard_pack_candidates: this rank's candidate entries as the blob of the candidate all-gather."""
    L = lib()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    out, nb = ctypes.c_void_p(), ctypes.c_int64(0)
    L.dh_shard_pack_candidates.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    L.dh_shard_free.argtypes = [ctypes.c_void_p]
    _check(L.dh_shard_pack_candidates(cands._h, arr.ctypes.data, len(arr), read_shift, ctypes.byref(out), ctypes.byref(nb)))
    b = _blob(out, nb.value)
    L.dh_shard_free(out)
    return b


def shard_read_joins(las, contig_off, read_off, read_first=0):
    """dh_shard_read_joins: the raw scaffold joins of this rank's reads (global read ids in las["bread"], read_off =
    offsets of the rank's reads [read_first, read_first + len(read_off) - 1)) as the blob of the join all-gather."""
    L = lib()
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    co, ro = np.ascontiguousarray(contig_off, dtype=np.int64), np.ascontiguousarray(read_off, dtype=np.int64)
    out, nb = ctypes.c_void_p(), ctypes.c_int64(0)
    L.dh_shard_read_joins.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    L.dh_shard_free.argtypes = [ctypes.c_void_p]
    _check(L.dh_shard_read_joins(arr.ctypes.data, len(arr), co.ctypes.data, len(co) - 1, ro.ctypes.data, int(read_first),
                                 len(ro) - 1, ctypes.byref(out), ctypes.byref(nb)))
    b = _blob(out, nb.value)
    L.dh_shard_free(out)
    return b


class ShardPlan:
    """dh_shard_plan: the pile-ups every rank derives from the gathered candidates (or, graph = (ncontigs, input_gaps,
    scaffold options): from the gathered scaffold joins), and who processes them."""

    def __init__(self, blobs, opts, graph=None):
        L = lib()
        self._blobs = [np.ascontiguousarray(b, dtype=np.uint8) for b in blobs]
        n = len(self._blobs)
        ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in self._blobs])
        sizes = (ctypes.c_int64 * n)(*[len(b) for b in self._blobs])
        h = ctypes.c_void_p()
        if graph is not None:
            ncontigs, input_gaps, kw = graph
            so = ScaffoldOpts()
            L.dh_default_scaffold_opts(ctypes.byref(so))
            for k, v in (kw or {}).items():
                if not hasattr(so, k):
                    raise TypeError(f"unknown scaffold option {k}")
                setattr(so, k, int(v) if k in _SCAFFOLD_INT_OPTS else float(v))
            ig = np.ascontiguousarray(input_gaps if input_gaps is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
            L.dh_shard_graph_plan_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                     ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ScaffoldOpts),
                                                     ctypes.POINTER(ProcessOpts), ctypes.POINTER(ctypes.c_void_p)]
            _check(L.dh_shard_graph_plan_create(ptrs, sizes, n, int(ncontigs), ig.ctypes.data if len(ig) else None, len(ig),
                                                ctypes.byref(so), ctypes.byref(opts), ctypes.byref(h)))
        else:
            L.dh_shard_plan_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ProcessOpts),
                                               ctypes.POINTER(ctypes.c_void_p)]
            _check(L.dh_shard_plan_create(ptrs, sizes, n, ctypes.byref(opts), ctypes.byref(h)))
        self._h = h
        for fn in (L.dh_shard_plan_las, L.dh_shard_plan_pileups, L.dh_shard_plan_owner):
            fn.argtypes = [ctypes.c_void_p]
            fn.restype = ctypes.c_void_p
        L.dh_shard_plan_nlas.argtypes = [ctypes.c_void_p]
        L.dh_shard_plan_nlas.restype = ctypes.c_int64
        nl = L.dh_shard_plan_nlas(h)
        # a view of the plan's records (kept alive by this object)
        self.las = (np.frombuffer((ctypes.c_uint8 * (nl * LA_DTYPE.itemsize)).from_address(L.dh_shard_plan_las(h)), dtype=LA_DTYPE)
                    if nl else np.zeros(0, dtype=LA_DTYPE))
        self.piles = Pileups(None, None, None, _handle=ctypes.c_void_p(L.dh_shard_plan_pileups(h)), _keep=self)
        npl = len(self.piles)
        self.owner = (np.frombuffer((ctypes.c_uint8 * (4 * npl)).from_address(L.dh_shard_plan_owner(h)), dtype=np.int32).copy()
                      if npl else np.zeros(0, np.int32))

    def close(self):
        if self._h:
            self.piles._h = None
            self.las = None
            L = lib()
            L.dh_shard_plan_destroy.argtypes = [ctypes.c_void_p]
            L.dh_shard_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """dh_comm: the communicator of the sharded path behind the C ABI -- RCCL between processes (Comm.create) or the
    in-process hub between host threads (Comm.local)."""

    def __init__(self, h):
        self._h = h

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * 128)()
        _check(lib().dh_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def create(ctx, rank, world, uid):
        L = lib()
        L.dh_comm_create.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        h = ctypes.c_void_p()
        _check(L.dh_comm_create(uid, rank, world, ctx._h, ctypes.byref(h)))
        return Comm(h)

    @staticmethod
    def local(world, ctxs=None):
        L = lib()
        hs = (ctypes.c_void_p * world)()
        cp = (ctypes.c_void_p * world)(*[c._h for c in ctxs]) if ctxs is not None else None
        L.dh_comm_create_local.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        _check(L.dh_comm_create_local(world, cp, hs))
        return [Comm(ctypes.c_void_p(hs[r])) for r in range(world)]

    @property
    def world(self):
        L = lib()
        L.dh_comm_world.argtypes = [ctypes.c_void_p]
        return int(L.dh_comm_world(self._h))

    def _take(self, ptr, sizes):
        L = lib()
        L.dh_shard_free.argtypes = [ctypes.c_void_p]
        tot = int(sum(sizes))
        # (ctypes.string_at takes a C int: blocks beyond 2 GB go through a typed view of the address)
        whole = np.frombuffer((ctypes.c_uint8 * tot).from_address(ptr.value), dtype=np.uint8).copy() if tot else np.zeros(0, np.uint8)
        L.dh_shard_free(ptr)
        out, at = [], 0
        for n in sizes:
            out.append(whole[at:at + int(n)].copy())
            at += int(n)
        return out

    def all_gather(self, payload):
        L = lib()
        p = np.ascontiguousarray(payload, dtype=np.uint8)
        W = self.world
        sizes = (ctypes.c_int64 * W)()
        out = ctypes.c_void_p()
        L.dh_comm_all_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        _check(L.dh_comm_all_gather(self._h, p.ctypes.data if len(p) else None, len(p), ctypes.byref(out), sizes))
        return self._take(out, list(sizes))

    def all_to_all(self, per_dest):
        L = lib()
        W = self.world
        bl = [np.ascontiguousarray(x, dtype=np.uint8) for x in per_dest]
        ptrs = (ctypes.c_void_p * W)(*[b.ctypes.data if len(b) else None for b in bl])
        ss = (ctypes.c_int64 * W)(*[len(b) for b in bl])
        rs = (ctypes.c_int64 * W)()
        out = ctypes.c_void_p()
        L.dh_comm_all_to_all.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        _check(L.dh_comm_all_to_all(self._h, ptrs, ss, ctypes.byref(out), rs))
        return self._take(out, list(rs))

    def close(self):
        if self._h:
            L = lib()
            L.dh_comm_destroy.argtypes = [ctypes.c_void_p]
            L.dh_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_run_prepare(comm, contigs_db, reads_db, read_first, contig_off, las, trace, opts, cands=None, graph=None):
    """Everything of shard_run that can fail BEFORE the C call -- array conversion, option names, missing handles -- done
    up front; returns the closure that makes the call.  dentist_amd.parallel lets the ranks agree that all of them got this
    far before any of them enters dh_shard_run's first exchange."""
    L = lib()
    for what, h in (("communicator", comm), ("contigs DB", contigs_db), ("reads DB", reads_db)):
        if h is None or not getattr(h, "_h", None):
            raise ValueError(f"shard_run: no {what}")
    if cands is None and graph is None:
        raise ValueError("shard_run: neither candidates nor graph options")
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    co = np.ascontiguousarray(contig_off, dtype=np.int64)
    if co.ndim != 1 or len(co) < 1:
        raise ValueError("shard_run: contig_off must hold at least one offset")
    ro = ig = so = None
    if cands is None:
        g = dict(graph or {})
        ro = np.ascontiguousarray(g.pop("read_off"), dtype=np.int64)
        gaps = g.pop("input_gaps", None)
        ig = np.ascontiguousarray(gaps if gaps is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
        so = ScaffoldOpts()
        L.dh_default_scaffold_opts(ctypes.byref(so))
        g.setdefault("min_spanning_reads", opts.min_reads)
        for k, v in g.items():
            if not hasattr(so, k):
                raise TypeError(f"unknown scaffold option {k}")
            setattr(so, k, int(v) if k in _SCAFFOLD_INT_OPTS else float(v))
    elif not getattr(cands, "_h", None):
        raise ValueError("shard_run: the candidates have no handle")
    vp = ctypes.c_void_p
    L.dh_shard_run.argtypes = [vp, vp, vp, ctypes.c_int32, vp, ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.POINTER(ProcessOpts),
                               vp, vp, vp, ctypes.c_int32, vp, ctypes.POINTER(vp), vp]

    def call():
        h = ctypes.c_void_p()
        info = (ctypes.c_int64 * 4)()
        _check(L.dh_shard_run(comm._h, contigs_db._h, reads_db._h, read_first, co.ctypes.data, len(co) - 1, arr.ctypes.data, len(arr),
                              tr.ctypes.data, ctypes.byref(opts), cands._h if cands is not None else None,
                              ro.ctypes.data if ro is not None else None, ig.ctypes.data if ig is not None and len(ig) else None,
                              len(ig) if ig is not None else 0, ctypes.byref(so) if so is not None else None, ctypes.byref(h), info))
        rec, bases = _take_insertions(h, None, False)
        return rec, bases, {"piles": int(info[0]), "owned": int(info[1]), "entries": int(info[2]), "cropped_bytes_sent": int(info[3])}
    return call


def shard_run(comm, contigs_db, reads_db, read_first, contig_off, las, trace, opts, cands=None, graph=None):
    """dh_shard_run: `collect` + `process` of one rank's share with its three exchanges behind the C ABI.  graph =
    dict(read_off=..., input_gaps=..., scaffold options) selects the scaffold-graph collector, cands the spanning-read
    one.  Returns (records of all ranks by gap, consensus bases, info)."""
    return shard_run_prepare(comm, contigs_db, reads_db, read_first, contig_off, las, trace, opts, cands=cands, graph=graph)()


def shard_pack_cropped(crop, owner, world):
    """dh_shard_pack_cropped: the reads this rank cropped, one blob per owner rank."""
    L = lib()
    ow = np.ascontiguousarray(owner, dtype=np.int32)
    ptrs = (ctypes.c_void_p * world)()
    sizes = (ctypes.c_int64 * world)()
    L.dh_shard_pack_cropped.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.dh_shard_free.argtypes = [ctypes.c_void_p]
    _check(L.dh_shard_pack_cropped(crop._h, ow.ctypes.data, world, ptrs, sizes))

    class _Block:   # the one malloc'd block behind all the blobs: released with the last view
        def __init__(self, p):
            self.p = p

        def __del__(self):
            try:
                lib().dh_shard_free(self.p)
            except Exception:
                pass
    total = sum(int(sizes[r]) for r in range(world))
    block = _Block(ptrs[0])
    whole = np.frombuffer((ctypes.c_uint8 * max(total, 1)).from_address(ptrs[0]), dtype=np.uint8)
    out, at = [], 0
    for r in range(world):   # views, no copy: the collective (or torch.from_numpy) reads them in place
        v = whole[at:at + int(sizes[r])]
        at += int(sizes[r])
        out.append(v)
    _KEEP[id(whole)] = block
    import weakref
    weakref.finalize(whole, _KEEP.pop, id(whole), None)
    return out


def shard_unpack_cropped(blobs, rec, owner, rank):
    """dh_shard_unpack_cropped: what an owner received -> Cropped for dh_process_cropped."""
    L = lib()
    bl = [np.ascontiguousarray(b, dtype=np.uint8) for b in blobs]
    n = len(bl)
    ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bl])
    sizes = (ctypes.c_int64 * n)(*[len(b) for b in bl])
    r = np.ascontiguousarray(rec, dtype=INSERTION_DTYPE)
    ow = np.ascontiguousarray(owner, dtype=np.int32)
    h = ctypes.c_void_p()
    L.dh_shard_unpack_cropped.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    _check(L.dh_shard_unpack_cropped(ptrs, sizes, n, r.ctypes.data, len(r), ow.ctypes.data, rank, ctypes.byref(h)))
    return Cropped(h)


def collect_filter(las, contig_off, read_off, opts, repeat_mask=None, inplace=False):
    """The six filters of `dentist collect` (filter.d:122-356). Returns (las with DISABLED flags -- a
    copy unless inplace, dropped-per-stage int64[6], read_used uint8[nreads])."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    if not inplace or arr is not las or not arr.flags.writeable:
        arr = arr.copy()
    co, ro = np.ascontiguousarray(contig_off, dtype=np.int64), np.ascontiguousarray(read_off, dtype=np.int64)
    dropped = np.zeros(6, dtype=np.int64)
    used = np.ones(len(ro) - 1, dtype=np.uint8)
    rp = ri = None
    if repeat_mask is not None:
        rp = np.ascontiguousarray(repeat_mask[0], dtype=np.int64)
        ri = np.ascontiguousarray(np.concatenate([repeat_mask[1], [0, 0]]), dtype=np.int32)
    L = lib()
    L.dh_collect_filter.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ProcessOpts),
                                    ctypes.c_void_p, ctypes.c_void_p]
    _check(L.dh_collect_filter(arr.ctypes.data, len(arr), co.ctypes.data, len(co) - 1, ro.ctypes.data, len(ro) - 1,
                               rp.ctypes.data if rp is not None else None, ri.ctypes.data if ri is not None else None,
                               ctypes.byref(opts), dropped.ctypes.data, used.ctypes.data))
    return arr, dropped, used


def propagate_mask(las, trace, tspace, mask, ncontigs, read_off):
    """dh_propagate_mask: the contig mask (ptr, iv) carried to the reads (propagateMask.d:136-305).
    Returns (ptr int64[nreads + 1], iv int32[m, 2])."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    mp = np.ascontiguousarray(mask[0], dtype=np.int64)
    mi = np.ascontiguousarray(np.concatenate([np.asarray(mask[1], dtype=np.int32).reshape(-1), [0, 0]]), dtype=np.int32)
    ro = np.ascontiguousarray(read_off, dtype=np.int64)
    L = lib()
    L.dh_propagate_mask.restype = ctypes.c_int64
    L.dh_propagate_mask.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int64]
    ptr = np.zeros(len(ro), dtype=np.int64)
    args = (arr.ctypes.data if len(arr) else None, len(arr), tr.ctypes.data if len(tr) else None, int(tspace),
            mp.ctypes.data, mi.ctypes.data, int(ncontigs), ro.ctypes.data, len(ro) - 1, ptr.ctypes.data)
    m = L.dh_propagate_mask(*args, None, 0)
    if m < 0:
        _check(int(m))
    iv = np.zeros((max(int(m), 1), 2), dtype=np.int32)
    L.dh_propagate_mask(*args, iv.ctypes.data, int(m))
    return ptr, iv[:int(m)]


REGION_DTYPE = np.dtype([("contig", "<i4"), ("begin", "<i4"), ("end", "<i4")])
REGION_REPORT_DTYPE = np.dtype([("num_spanning_reads", "<i4"), ("weak_bp", "<i4"), ("is_valid", "<i4"),
                                ("ctx_begin", "<i4"), ("ctx_end", "<i4")])


def validate_regions(las, contig_off, regions, min_coverage_reads, min_spanning_reads=3, region_context=1000,
                     weak_coverage_window=500):
    """dh_validate_regions (`dentist validate-regions`, validateRegions.d:325-512).  regions: (contig,
    begin, end) rows.  Returns (reports REGION_REPORT_DTYPE[nregions], weak (contig, begin, end) int32[m, 3])."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    co = np.ascontiguousarray(contig_off, dtype=np.int64)
    rg = np.zeros(len(regions), dtype=REGION_DTYPE)
    r = np.asarray(regions, dtype=np.int32).reshape(-1, 3)
    rg["contig"], rg["begin"], rg["end"] = r[:, 0], r[:, 1], r[:, 2]
    rep = np.zeros(len(rg), dtype=REGION_REPORT_DTYPE)
    L = lib()
    L.dh_validate_regions.restype = ctypes.c_int64
    L.dh_validate_regions.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    args = (arr.ctypes.data if len(arr) else None, len(arr), co.ctypes.data, len(co) - 1, rg.ctypes.data if len(rg) else None,
            len(rg), int(region_context), int(weak_coverage_window), int(min_coverage_reads), int(min_spanning_reads),
            rep.ctypes.data if len(rg) else None)
    m = L.dh_validate_regions(*args, None, 0)
    if m < 0:
        _check(int(m))
    weak = np.zeros((max(int(m), 1), 3), dtype=np.int32)
    L.dh_validate_regions(*args, weak.ctypes.data, int(m))
    return rep, weak[:int(m)]


def max_coverage_reads(read_coverage):
    """--max-coverage-reads derived from --read-coverage (commandline.d:1876-1889)."""
    L = lib()
    L.dh_max_coverage_reads.argtypes = [ctypes.c_double]
    L.dh_max_coverage_reads.restype = ctypes.c_int32
    return int(L.dh_max_coverage_reads(float(read_coverage)))


def max_improper_coverage_reads(read_coverage):
    """--max-improper-coverage-reads derived from --read-coverage (commandline.d:1957-1970)."""
    L = lib()
    L.dh_max_improper_coverage_reads.argtypes = [ctypes.c_double]
    L.dh_max_improper_coverage_reads.restype = ctypes.c_int32
    return int(L.dh_max_improper_coverage_reads(float(read_coverage)))


class ScaffoldOpts(ctypes.Structure):
    _fields_ = [("min_spanning_reads", ctypes.c_int32), ("merge_extensions", ctypes.c_int32),
                ("best_pile_up_margin", ctypes.c_double), ("existing_gap_bonus", ctypes.c_double),
                ("only_joins", ctypes.c_int32), ("pad_", ctypes.c_int32)]


_SCAFFOLD_INT_OPTS = ("min_spanning_reads", "merge_extensions", "only_joins")


JOIN_DTYPE = np.dtype([("contig0", "<i4"), ("part0", "<i4"), ("contig1", "<i4"), ("part1", "<i4"), ("type", "<i4"),
                       ("count", "<i4"), ("first", "<i8")])
READ_ALIGNMENT_DTYPE = np.dtype([("read", "<i4"), ("la0", "<i4"), ("la1", "<i4"), ("seed0", "u1"), ("seed1", "u1"),
                                 ("n", "u1"), ("pad", "u1")])


REMAP_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                            ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p),
                            ctypes.POINTER(ctypes.c_int64))


def _scaffold(las, contig_off, read_off, input_gaps, kw, resolve=None):
    """The scaffold handle and the LA array it indexes.  resolve = None: dh_scaffold_pileups.  resolve = dict(ctx=, contigs=,
    reads=, map_opts=, trace=, allowance=100): dh_scaffold_pileups_resolved (resolveBubbles, the device re-maps the skipping
    reads); resolve = dict(remap=callable(contig_ids, read_ids) -> LA_DTYPE array): dh_scaffold_pileups_cb.  With resolve the
    result carries (las incl. the added alignments, trace incl. theirs or None, bubbles resolved) in the handle's slot 2."""
    L = lib()
    o = ScaffoldOpts()
    L.dh_default_scaffold_opts(ctypes.byref(o))
    kw = dict(kw)
    mbs, mit = int(kw.pop("max_bubble_size", 0)), int(kw.pop("max_bubble_iterations", 0))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown scaffold option {k}")
        setattr(o, k, int(v) if k in ("min_spanning_reads", "merge_extensions") else float(v))
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    ig = np.ascontiguousarray(input_gaps if input_gaps is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
    h = ctypes.c_void_p()
    vp = ctypes.c_void_p
    if resolve is None:
        co, ro = np.ascontiguousarray(contig_off, dtype=np.int64), np.ascontiguousarray(read_off, dtype=np.int64)
        L.dh_scaffold_pileups.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int32, vp, ctypes.c_int32, vp, ctypes.c_int32,
                                          ctypes.POINTER(ScaffoldOpts), ctypes.POINTER(vp)]
        _check(L.dh_scaffold_pileups(arr.ctypes.data, len(arr), co.ctypes.data, len(co) - 1, ro.ctypes.data, len(ro) - 1,
                                     ig.ctypes.data if len(ig) else None, len(ig), ctypes.byref(o), ctypes.byref(h)))
        return h, arr
    ex = vp()
    nres = ctypes.c_int32(0)
    if "remap" in resolve:
        co, ro = np.ascontiguousarray(contig_off, dtype=np.int64), np.ascontiguousarray(read_off, dtype=np.int64)
        fn = resolve["remap"]
        libc = ctypes.CDLL(None)
        libc.malloc.restype = vp
        libc.malloc.argtypes = [ctypes.c_size_t]

        def cb(user, cids, nc, rids, nr, out, nout):
            try:
                got = np.ascontiguousarray(fn([cids[i] for i in range(nc)], [rids[i] for i in range(nr)]), dtype=LA_DTYPE)
                p = libc.malloc(max(1, got.nbytes))
                if got.nbytes:
                    ctypes.memmove(p, got.ctypes.data, got.nbytes)
                out[0] = p
                nout[0] = len(got)
                return 0
            except Exception:  # noqa: BLE001
                return -1
        cfn = REMAP_FN(cb)
        L.dh_scaffold_pileups_cb.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int32, vp, ctypes.c_int32, vp, ctypes.c_int32,
                                             ctypes.POINTER(ScaffoldOpts), ctypes.c_int32, ctypes.c_int32, REMAP_FN, vp,
                                             ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
        _check(L.dh_scaffold_pileups_cb(arr.ctypes.data, len(arr), co.ctypes.data, len(co) - 1, ro.ctypes.data, len(ro) - 1,
                                        ig.ctypes.data if len(ig) else None, len(ig), ctypes.byref(o), mbs, mit, cfn, None,
                                        ctypes.byref(h), ctypes.byref(ex), ctypes.byref(nres)))
        trace = None
    else:
        L.dh_scaffold_pileups_resolved.argtypes = [vp, vp, vp, vp, ctypes.c_int64, vp, ctypes.c_int32, ctypes.POINTER(ScaffoldOpts),
                                                   ctypes.POINTER(AlignOpts), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
        _check(L.dh_scaffold_pileups_resolved(resolve["ctx"]._h, resolve["contigs"]._h, resolve["reads"]._h, arr.ctypes.data, len(arr),
                                              ig.ctypes.data if len(ig) else None, len(ig), ctypes.byref(o),
                                              ctypes.byref(resolve["map_opts"]), int(resolve.get("allowance", 100)), mbs, mit,
                                              ctypes.byref(h), ctypes.byref(ex), ctypes.byref(nres)))
        trace = np.ascontiguousarray(resolve["trace"], dtype=np.uint16)
    xl, xt, _ = _take_la_set(ex)
    if len(xl):
        xl = xl.copy()
        if trace is not None:
            xl["toff"] += len(trace)
            trace = np.concatenate([trace, xt])
        arr = np.concatenate([arr, xl])
    return h, arr, trace, int(nres.value)


def scaffold_pileups(las, contig_off, read_off, input_gaps=None, resolve=None, **opts):
    """dh_scaffold_pileups: the scaffold-graph pile-up builder of `dentist collect`
    (pileups.d:173-208).  Returns (joins JOIN_DTYPE[npiles], read alignments READ_ALIGNMENT_DTYPE[...]); with
    resolve (see _scaffold: resolveBubbles) also (las incl. the added alignments, trace or None, bubbles resolved)."""
    L = lib()
    sc = _scaffold(las, contig_off, read_off, input_gaps, opts, resolve)
    h = sc[0]
    try:
        L.dh_scaffold_npiles.restype = ctypes.c_int32
        L.dh_scaffold_nentries.restype = ctypes.c_int64
        L.dh_scaffold_joins.restype = ctypes.c_void_p
        L.dh_scaffold_entries.restype = ctypes.c_void_p
        for f in (L.dh_scaffold_npiles, L.dh_scaffold_nentries, L.dh_scaffold_joins, L.dh_scaffold_entries):
            f.argtypes = [ctypes.c_void_p]
        nj, ne = L.dh_scaffold_npiles(h), L.dh_scaffold_nentries(h)
        joins = np.zeros(nj, dtype=JOIN_DTYPE)
        ent = np.zeros(ne, dtype=READ_ALIGNMENT_DTYPE)
        if nj:
            ctypes.memmove(joins.ctypes.data, L.dh_scaffold_joins(h), joins.nbytes)
        if ne:
            ctypes.memmove(ent.ctypes.data, L.dh_scaffold_entries(h), ent.nbytes)
    finally:
        L.dh_scaffold_destroy.argtypes = [ctypes.c_void_p]
        L.dh_scaffold_destroy(h)
    return (joins, ent) if resolve is None else (joins, ent, sc[1], sc[2], sc[3])


def scaffold_spanning_pileups(las, contig_off, read_off, input_gaps=None, with_extensions=False, resolve=None, **opts):
    """The gap pile-ups of the scaffold graph (collectPileUps/pileups.d:173-208): returns (Pileups, number of
    pile-ups of other kinds).  with_extensions = False: the spanning reads only (dh_scaffold_spanning);
    True: every read alignment of the pile-up, i.e. also the extension entries mergeExtensionsWithGaps moved
    into the gap, as (read, LA, -1) / (read, -1, LA) triples (dh_scaffold_gap_pileups) -- what the reference's
    `dentist process` is handed."""
    L = lib()
    sc = _scaffold(las, contig_off, read_off, input_gaps, opts, resolve)
    h, arr = sc[0], sc[1]
    try:
        ph = ctypes.c_void_p()
        skipped = ctypes.c_int32(0)
        fn = L.dh_scaffold_gap_pileups if with_extensions else L.dh_scaffold_spanning
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32)]
        _check(fn(h, arr.ctypes.data, len(arr), ctypes.byref(ph), ctypes.byref(skipped)))
    finally:
        L.dh_scaffold_destroy.argtypes = [ctypes.c_void_p]
        L.dh_scaffold_destroy(h)
    if resolve is not None:   # + the LA / trace arrays the pile-ups index, bubbles resolved
        return Pileups(None, None, None, _handle=ph), int(skipped.value), sc[1], sc[2], sc[3]
    return Pileups(None, None, None, _handle=ph), int(skipped.value)


def scaffold_all_pileups(las, contig_off, read_off, input_gaps=None, only="both", resolve=None, **opts):
    """Every pile-up of the scaffold graph the process stage can take (dh_scaffold_all_pileups): only = "spanning" (gap
    joins of any two contig ends), "extending" (extension joins) or "both".  Returns (Pileups with joins, skipped)."""
    L = lib()
    sc = _scaffold(las, contig_off, read_off, input_gaps, opts, resolve)
    h, arr = sc[0], sc[1]
    try:
        ph = ctypes.c_void_p()
        skipped = ctypes.c_int32(0)
        L.dh_scaffold_all_pileups.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32)]
        _check(L.dh_scaffold_all_pileups(h, arr.ctypes.data, len(arr), {"spanning": 1, "extending": 2, "both": 3}[only],
                                         ctypes.byref(ph), ctypes.byref(skipped)))
    finally:
        L.dh_scaffold_destroy.argtypes = [ctypes.c_void_p]
        L.dh_scaffold_destroy(h)
    if resolve is not None:
        return Pileups(None, None, None, _handle=ph), int(skipped.value), sc[1], sc[2], sc[3]
    return Pileups(None, None, None, _handle=ph), int(skipped.value)


class Cropped:
    """Cropped pile-ups (dh_cropped): per pile-up record + per cropped read (pile, entry, read id, bases)."""

    def __init__(self, h):
        self._h = h

    @classmethod
    def crop(cls, ctx, contigs, reads, read_first, las, trace, piles, opts, repeat_mask=None):
        arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
        tr = np.ascontiguousarray(trace, dtype=np.uint16)
        h = ctypes.c_void_p()
        keep, rp, ri = _mask_args(repeat_mask)
        L = lib()
        L.dh_crop_pileups_masked.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 6
        _check(L.dh_crop_pileups_masked(ctx._h, contigs._h, reads._h, read_first, arr.ctypes.data, len(arr),
                                        tr.ctypes.data if len(tr) else None, piles._h, rp, ri, ctypes.byref(opts), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def create(cls, rec, pile, entry, read_id, off, bases, kind=None):
        r = np.ascontiguousarray(rec, dtype=INSERTION_DTYPE)
        pi, en, ri = (np.ascontiguousarray(x, dtype=np.int32) for x in (pile, entry, read_id))
        of = np.ascontiguousarray(off, dtype=np.int64)
        ba = np.ascontiguousarray(bases, dtype=np.uint8)
        ki = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
        h = ctypes.c_void_p()
        _check(lib().dh_cropped_create2(r.ctypes.data, len(r), len(pi), pi.ctypes.data, en.ctypes.data, ri.ctypes.data,
                                        ki.ctypes.data if ki is not None and len(ki) else None,
                                        of.ctypes.data, ba.ctypes.data if len(ba) else None, ctypes.byref(h)))
        return cls(h)

    def kind(self):
        """Per cropped read: 0 spans the gap, 1 / 2 extension entry of the left / right contig."""
        L, h = lib(), self._h
        nr = L.dh_cropped_nreads(h)
        if not nr:
            return np.zeros(0, np.uint8)
        buf = (ctypes.c_uint8 * nr).from_address(L.dh_cropped_kind(h))
        return np.frombuffer(buf, dtype=np.uint8).copy()

    def arrays(self):
        """(records, pile, entry, read_id, off, bases) as numpy copies (bases come over PCIe)."""
        L, h = lib(), self._h
        npl, nr = L.dh_cropped_npiles(h), L.dh_cropped_nreads(h)

        def arr(ptr, n, dt):  # one copy out of the library's buffer
            if not n:
                return np.zeros(0, dt)
            buf = (ctypes.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).copy()
        rec = arr(L.dh_cropped_records(h), npl, INSERTION_DTYPE)
        off = arr(L.dh_cropped_offsets(h), nr + 1, np.int64) if nr else np.zeros(1, np.int64)
        bp = L.dh_cropped_bases(h)
        if not bp:
            _check(-3)
        return (rec, arr(L.dh_cropped_pile(h), nr, np.int32), arr(L.dh_cropped_entry(h), nr, np.int32),
                arr(L.dh_cropped_read_id(h), nr, np.int32), off, arr(bp, int(off[-1]), np.uint8))

    def process(self, ctx, contigs, opts):
        h = ctypes.c_void_p()
        _check(lib().dh_process_cropped(ctx._h, contigs._h, self._h, ctypes.byref(opts), ctypes.byref(h)))
        return _take_insertions(h)

    def close(self):
        if self._h:
            lib().dh_cropped_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take_insertions(h, insertions_db=None, read_ids=False):
    """insertions_db = (path, contig_off, tspace): also write the result as DENTIST's insertions.db.
    read_ids: also return (ids, off): the 0-based read ids of every record's pile-up."""
    L = lib()
    if insertions_db is not None:
        path, contig_off, tspace = insertions_db
        off = np.ascontiguousarray(contig_off, dtype=np.int64)
        L.dh_insertions_write_db.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p]
        rc = L.dh_insertions_write_db(h, off.ctypes.data, len(off) - 1, tspace, path.encode())
        if rc != 0:
            L.dh_insertions_destroy(h)
            _check(rc)
    n = L.dh_insertions_count(h)
    nb = L.dh_insertions_bases_len(h)
    rec = (np.frombuffer(ctypes.string_at(L.dh_insertions_records(h), n * INSERTION_DTYPE.itemsize),
                         dtype=INSERTION_DTYPE).copy() if n else np.zeros(0, dtype=INSERTION_DTYPE))
    bases = (np.frombuffer(ctypes.string_at(L.dh_insertions_bases(h), nb), dtype=np.uint8).copy()
             if nb else np.zeros(0, dtype=np.uint8))
    ids = None
    if read_ids:
        L.dh_insertions_read_ids.restype = ctypes.c_void_p
        L.dh_insertions_read_ids_off.restype = ctypes.c_void_p
        L.dh_insertions_read_ids.argtypes = L.dh_insertions_read_ids_off.argtypes = [ctypes.c_void_p]
        po = L.dh_insertions_read_ids_off(h)
        if po and n:
            off = np.frombuffer(ctypes.string_at(po, 4 * (n + 1)), dtype=np.int32).astype(np.int64)
            tot = int(off[-1])
            idv = (np.frombuffer(ctypes.string_at(L.dh_insertions_read_ids(h), 4 * tot), dtype=np.int32).copy()
                   if tot else np.zeros(0, np.int32))
            ids = (idv, off)
        else:
            ids = (np.zeros(0, np.int32), np.zeros(n + 1, np.int64))
    L.dh_insertions_destroy(h)
    return (rec, bases, ids) if read_ids else (rec, bases)


def _mask_args(repeat_mask):
    """(ptr int64[ncontigs + 1], intervals int32[2 * total]) -> the two pointers of the *_masked entries (None, None without a mask)."""
    if repeat_mask is None:
        return None, None, None
    rp = np.ascontiguousarray(repeat_mask[0], dtype=np.int64)
    ri = np.ascontiguousarray(np.concatenate([np.asarray(repeat_mask[1]).reshape(-1), [0, 0]]), dtype=np.int32)
    return (rp, ri), rp.ctypes.data, ri.ctypes.data


def process_pileups(ctx, contigs, reads, las, trace, piles, opts, insertions_db=None, read_ids=False, repeat_mask=None):
    """dentist `process` for a batch of pile-ups on the GPU. Returns (records, consensus bases);
    insertions_db = (path, contig_off, tspace) also writes DENTIST's insertions.db; read_ids = True adds
    (ids, off): the read ids of every record's pile-up (what `dentist output` lists in its BED / AGP);
    repeat_mask = (ptr, intervals): the contigs' repeat mask for the cropper (dh_process_pileups_masked)."""
    L = lib()
    h = ctypes.c_void_p()
    keep, rp, ri = _mask_args(repeat_mask)
    if isinstance(trace, DeviceTrace):   # (the records are the set's own: `las` is the view of them map_reads returned)
        L.dh_process_pileups_set.argtypes = [ctypes.c_void_p] * 9
        _check(L.dh_process_pileups_set(ctx._h, contigs._h, reads._h, trace._h, piles._h, rp, ri, ctypes.byref(opts), ctypes.byref(h)))
        return _take_insertions(h, insertions_db, read_ids)
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    L.dh_process_pileups_masked.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_void_p] * 6
    _check(L.dh_process_pileups_masked(ctx._h, contigs._h, reads._h, arr.ctypes.data, len(arr), tr.ctypes.data,
                                       piles._h, rp, ri, ctypes.byref(opts), ctypes.byref(h)))
    return _take_insertions(h, insertions_db, read_ids)


class OutputOpts(ctypes.Structure):
    _fields_ = [("line_width", ctypes.c_int32), ("highlight", ctypes.c_int32), ("join_policy", ctypes.c_int32),
                ("agp_dazzler", ctypes.c_int32), ("agp_skip_read_ids", ctypes.c_int32), ("only", ctypes.c_int32),
                ("agp_version", ctypes.c_char_p), ("tool", ctypes.c_char_p), ("input_assembly", ctypes.c_char_p),
                ("min_extension_length", ctypes.c_int32), ("pad", ctypes.c_int32)]


JOIN_POLICIES = {"scaffoldGaps": 0, "scaffolds": 1, "contigs": 2}


def output_assembly(fasta_path, contigs, scaffold_of, headers, gap_len, rec, bases, read_ids=None, bed_path=None,
                    agp_path=None, join_policy="scaffoldGaps", agp_dazzler=False, agp_skip_read_ids=False, read_names=None,
                    line_width=50, highlight=True, tool=None, input_assembly=None, only="spanning", min_extension_length=100):
    """`dentist output` (host only): assembly graph with the join policy, fixCropping, FASTA, AGP and closed-gaps
    BED (dh_output_assembly).  read_ids = (ids, off) from process_pileups(read_ids=True).  Returns the number of
    insertions the join policy dropped."""
    L = lib()
    o = OutputOpts()
    L.dh_default_output_opts.argtypes = [ctypes.POINTER(OutputOpts)]
    L.dh_default_output_opts(ctypes.byref(o))
    o.line_width, o.highlight, o.join_policy = line_width, int(bool(highlight)), JOIN_POLICIES[join_policy]
    o.agp_dazzler, o.agp_skip_read_ids = int(bool(agp_dazzler)), int(bool(agp_skip_read_ids))
    o.only, o.min_extension_length = {"spanning": 1, "extending": 2, "both": 3}[only], int(min_extension_length)
    if tool is not None:
        o.tool = tool.encode()
    if input_assembly is not None:
        o.input_assembly = input_assembly.encode()
    cb = np.ascontiguousarray(contigs.bases, dtype=np.uint8)
    co = np.ascontiguousarray(contigs.off, dtype=np.int64)
    so = np.ascontiguousarray(scaffold_of, dtype=np.int32)
    gl = np.ascontiguousarray(gap_len, dtype=np.int32) if gap_len is not None else None
    r = np.ascontiguousarray(rec, dtype=INSERTION_DTYPE)
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    hs = (ctypes.c_char_p * len(headers))(*[h.encode() for h in headers])
    ids = off = None
    if read_ids is not None:
        ids = np.ascontiguousarray(read_ids[0], dtype=np.int32)
        off = np.ascontiguousarray(read_ids[1], dtype=np.int32)
    rn = (ctypes.c_char_p * len(read_names))(*[x.encode() for x in read_names]) if read_names is not None else None
    dropped = ctypes.c_int32(0)
    vp = ctypes.c_void_p
    L.dh_output_assembly.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, vp, vp, ctypes.c_int32, vp, vp, vp, vp,
                                     ctypes.c_int32, vp, vp, vp, ctypes.c_int32, vp, ctypes.POINTER(OutputOpts), ctypes.POINTER(ctypes.c_int32)]
    _check(L.dh_output_assembly(fasta_path.encode(), bed_path.encode() if bed_path else None,
                                agp_path.encode() if agp_path else None, cb.ctypes.data, co.ctypes.data, len(co) - 1,
                                so.ctypes.data, ctypes.cast(hs, vp), gl.ctypes.data if gl is not None else None,
                                r.ctypes.data, len(r), b.ctypes.data if len(b) else None,
                                ids.ctypes.data if ids is not None and len(ids) else (ids.ctypes.data if ids is not None else None),
                                off.ctypes.data if off is not None else None, len(read_names) if read_names is not None else -1,
                                ctypes.cast(rn, vp) if rn is not None else None,
                                ctypes.byref(o), ctypes.byref(dropped)))
    return int(dropped.value)


def output_fasta(fasta_path, contigs, scaffold_of, headers, gap_len, rec, bases, bed_path=None, line_width=50,
                 highlight=True):
    """`dentist output` for linear scaffolds (host only): contigs = SeqDb-like (.bases, .off),
    scaffold_of[c] = input scaffold of contig c, headers[s] = its FASTA header without '>',
    gap_len[c] = gap after contig c; rec / bases = result of process_pileups."""
    cb = np.ascontiguousarray(contigs.bases, dtype=np.uint8)
    co = np.ascontiguousarray(contigs.off, dtype=np.int64)
    so = np.ascontiguousarray(scaffold_of, dtype=np.int32)
    gl = np.ascontiguousarray(gap_len, dtype=np.int32) if gap_len is not None else None
    r = np.ascontiguousarray(rec, dtype=INSERTION_DTYPE)
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    hs = (ctypes.c_char_p * len(headers))(*[h.encode() for h in headers])
    _check(lib().dh_output_fasta(fasta_path.encode(), bed_path.encode() if bed_path else None, cb.ctypes.data,
                                 co.ctypes.data, len(co) - 1, so.ctypes.data, hs,
                                 gl.ctypes.data if gl is not None else None, r.ctypes.data, len(r),
                                 b.ctypes.data if len(b) else None, line_width, int(bool(highlight))))


def las_merge(paths, out_path):
    """LAmerge: several .las files (same trace spacing) into one, LAsort order (host only)."""
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    _check(lib().dh_las_merge(arr, len(paths), out_path.encode()))


def tile_qv(ctx, db, las, trace, tspace, cov, maxtiles):
    """DAScover + DASqv: intrinsic QV per (read, tile) of a pile-up DB; las grouped by aread."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    qv = np.zeros((lib().dh_db_nreads(db._h), maxtiles), dtype=np.uint8)
    _check(lib().dh_tile_qv(ctx._h, db._h, arr.ctypes.data, len(arr), tr.ctypes.data, tspace, cov, qv.ctypes.data,
                            maxtiles))
    return qv


def consensus(ctx, db, las, trace, tspace, ref_read, rounds=1):
    """computeintrinsicqv + daccord: consensus of read ref_read of the DB from its overlaps."""
    arr = np.ascontiguousarray(las, dtype=LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    cap = 1 << 20
    while True:
        out = np.zeros(cap, dtype=np.uint8)
        n = ctypes.c_int64(0)
        rc = lib().dh_consensus(ctx._h, db._h, arr.ctypes.data, len(arr), tr.ctypes.data, tspace, ref_read, rounds,
                                out.ctypes.data, cap, ctypes.byref(n))
        if rc != 0 and n.value > cap:
            cap = int(n.value)
            continue
        _check(rc)
        return out[:n.value].copy()


def process_stats(ctx):
    ms = (ctypes.c_float * 7)()
    cnt = (ctypes.c_int64 * 3)()
    _check(lib().dh_get_process_stats(ctx._h, ms, cnt))
    names = ("crop", "pile_align", "tile_qv", "consensus", "realign", "flank_align", "total")
    d = {f"ms_{n}": float(ms[i]) for i, n in enumerate(names)}
    d.update(pile_las=int(cnt[0]), tiles=int(cnt[1]), nw_cells=int(cnt[2]))
    wk = (ctypes.c_int64 * 4)()
    _check(lib().dh_get_process_work(ctx._h, wk))
    d.update(pile_ups=int(wk[0]), entries=int(wk[1]), cropped_bases=int(wk[2]), algorithmic_bytes=int(wk[3]))
    return d


# ---------------------------------------------------------------- DAZZ_DB files (host only)
def dazz_create_dam(path, fasta_text):
    b = fasta_text.encode() if isinstance(fasta_text, str) else fasta_text
    _check(lib().dh_dazz_create_dam(path.encode(), b, len(b)))


def dazz_create_db(path, fasta_text):
    b = fasta_text.encode() if isinstance(fasta_text, str) else fasta_text
    _check(lib().dh_dazz_create_db(path.encode(), b, len(b)))


def dazz_split(path, cutoff=0, all_reads=True, size_mb=200):
    _check(lib().dh_dazz_split(path.encode(), cutoff, int(all_reads), size_mb))


class DazzDb:
    """Trimmed view of a DAZZ_DB (or of one block, e.g. ``reads.3``) read from disk."""

    def __init__(self, path):
        L = lib()
        h = ctypes.c_void_p()
        _check(L.dh_dazz_open(path.encode(), ctypes.byref(h)))
        n = L.dh_dazz_nreads(h)
        self.first_id = L.dh_dazz_first_id(h)
        self.off = np.frombuffer((ctypes.c_int64 * (n + 1)).from_address(L.dh_dazz_offsets(h)), dtype=np.int64).copy()
        tot = int(self.off[-1])
        self.bases = (np.frombuffer((ctypes.c_uint8 * tot).from_address(L.dh_dazz_bases(h)), dtype=np.uint8).copy()
                      if tot else np.zeros(0, np.uint8))
        self.origin = (np.frombuffer((ctypes.c_int32 * n).from_address(L.dh_dazz_origin(h)), dtype=np.int32).copy()
                       if n else np.zeros(0, np.int32))
        self.fpulse = (np.frombuffer((ctypes.c_int32 * n).from_address(L.dh_dazz_fpulse(h)), dtype=np.int32).copy()
                       if n else np.zeros(0, np.int32))
        self.headers = [L.dh_dazz_header(h, i).decode() for i in range(n)]
        self.group = None
        self.mask = None
        self._h = h
        self._path = path

    def read_mask(self, name):
        """Intervals of mask track `name` for this (trimmed) view: (ptr int64[n+1], iv int32 pairs)."""
        L = lib()
        ptr = np.zeros(self.n + 1, dtype=np.int64)
        m = L.dh_dazz_read_mask(self._h, self._path.encode(), name.encode(), ptr.ctypes.data, None, 0)
        if m < 0:
            _check(int(m))
        iv = np.zeros(max(2 * m, 2), dtype=np.int32)
        L.dh_dazz_read_mask(self._h, self._path.encode(), name.encode(), ptr.ctypes.data, iv.ctypes.data, m)
        return ptr, iv[:2 * m]

    def close(self):
        if getattr(self, "_h", None):
            lib().dh_dazz_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return len(self.off) - 1

    def seq(self, i):
        return self.bases[self.off[i]:self.off[i + 1]]


def dazz_write_mask(db_path, name, ptr, iv):
    p = np.ascontiguousarray(ptr, dtype=np.int64)
    v = np.ascontiguousarray(iv, dtype=np.int32)
    _check(lib().dh_dazz_write_mask(db_path.encode(), name.encode(), len(p) - 1, p.ctypes.data, v.ctypes.data))


# ---------------------------------------------------------------- pile-ups.db / insertions.db (host only)
SEEDED_DTYPE = np.dtype([("id", "<i8"), ("contig_a_id", "<u4"), ("contig_a_len", "<u4"), ("contig_b_id", "<u4"),
                         ("contig_b_len", "<u4"), ("flags", "u1"), ("seed", "u1"), ("tspace", "<u2"), ("nla", "<i4")])
CHAIN_LA_DTYPE = np.dtype([("a_begin", "<u4"), ("a_end", "<u4"), ("b_begin", "<u4"), ("b_end", "<u4"), ("diffs", "<u4"),
                           ("ntp", "<i4")])
INSERTION_REC_DTYPE = np.dtype([("start_contig", "<i8"), ("end_contig", "<i8"), ("start_part", "u1"), ("end_part", "u1"),
                                ("pad", "u1", (6,)), ("seq_len", "<i8"), ("contig_len", "<i8"), ("noverlaps", "<i4"),
                                ("nread_ids", "<i4")])
assert SEEDED_DTYPE.itemsize == 32 and CHAIN_LA_DTYPE.itemsize == 24 and INSERTION_REC_DTYPE.itemsize == 48


def _arr(ptr, n, dt):
    # (no ctypes.string_at: it takes a C int, and a result set of a whole reads DB can be beyond 2 GB)
    if not n:
        return np.zeros(0, dt)
    addr = ptr if isinstance(ptr, int) else ctypes.cast(ptr, ctypes.c_void_p).value
    return np.frombuffer((ctypes.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(addr), dtype=dt).copy()


def _take_chaindb(h):
    L = lib()
    vp = ctypes.c_void_p
    for fn in ("dh_chaindb_pile_counts", "dh_chaindb_read_alignment_counts", "dh_chaindb_seeded", "dh_chaindb_las",
               "dh_chaindb_trace", "dh_chaindb_insertions", "dh_chaindb_bases", "dh_chaindb_read_ids"):
        getattr(L, fn).restype = vp
        getattr(L, fn).argtypes = [vp]
    for fn in ("dh_chaindb_nseeded", "dh_chaindb_nlas", "dh_chaindb_ntrace"):
        getattr(L, fn).restype = ctypes.c_int64
        getattr(L, fn).argtypes = [vp]
    for fn in ("dh_chaindb_npiles", "dh_chaindb_nread_alignments", "dh_chaindb_ninsertions"):
        getattr(L, fn).argtypes = [vp]
    L.dh_chaindb_destroy.argtypes = [vp]
    ins = _arr(L.dh_chaindb_insertions(h), L.dh_chaindb_ninsertions(h), INSERTION_REC_DTYPE)
    out = dict(pile_counts=_arr(L.dh_chaindb_pile_counts(h), L.dh_chaindb_npiles(h), np.int32),
               ra_counts=_arr(L.dh_chaindb_read_alignment_counts(h), L.dh_chaindb_nread_alignments(h), np.int32),
               seeded=_arr(L.dh_chaindb_seeded(h), L.dh_chaindb_nseeded(h), SEEDED_DTYPE),
               las=_arr(L.dh_chaindb_las(h), L.dh_chaindb_nlas(h), CHAIN_LA_DTYPE),
               trace=_arr(L.dh_chaindb_trace(h), L.dh_chaindb_ntrace(h), np.uint16),
               insertions=ins, bases=_arr(L.dh_chaindb_bases(h), int(ins["seq_len"].sum()) if len(ins) else 0, np.uint8),
               read_ids=_arr(L.dh_chaindb_read_ids(h), int(ins["nread_ids"].sum()) if len(ins) else 0, np.uint32))
    L.dh_chaindb_destroy(h)
    return out


def pileupdb_write(path, pile_counts, ra_counts, seeded, las, trace):
    pc, rc = (np.ascontiguousarray(x, dtype=np.int32) for x in (pile_counts, ra_counts))
    sa, la = np.ascontiguousarray(seeded, dtype=SEEDED_DTYPE), np.ascontiguousarray(las, dtype=CHAIN_LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    L = lib()
    L.dh_pileupdb_write.argtypes = [ctypes.c_char_p, ctypes.c_int32] + [ctypes.c_void_p] * 5
    _check(L.dh_pileupdb_write(path.encode(), len(pc), pc.ctypes.data, rc.ctypes.data, sa.ctypes.data, la.ctypes.data,
                               tr.ctypes.data))


def pileupdb_read(path):
    h = ctypes.c_void_p()
    L = lib()
    L.dh_pileupdb_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    _check(L.dh_pileupdb_read(path.encode(), ctypes.byref(h)))
    return _take_chaindb(h)


def insertiondb_write(path, insertions, bases, read_ids, seeded, las, trace):
    ins = np.ascontiguousarray(insertions, dtype=INSERTION_REC_DTYPE)
    b, ids = np.ascontiguousarray(bases, dtype=np.uint8), np.ascontiguousarray(read_ids, dtype=np.uint32)
    sa, la = np.ascontiguousarray(seeded, dtype=SEEDED_DTYPE), np.ascontiguousarray(las, dtype=CHAIN_LA_DTYPE)
    tr = np.ascontiguousarray(trace, dtype=np.uint16)
    L = lib()
    L.dh_insertiondb_write.argtypes = [ctypes.c_char_p, ctypes.c_int32] + [ctypes.c_void_p] * 6
    _check(L.dh_insertiondb_write(path.encode(), len(ins), ins.ctypes.data, b.ctypes.data, ids.ctypes.data, sa.ctypes.data,
                                  la.ctypes.data, tr.ctypes.data))


def insertiondb_read(path):
    h = ctypes.c_void_p()
    L = lib()
    L.dh_insertiondb_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    _check(L.dh_insertiondb_read(path.encode(), ctypes.byref(h)))
    return _take_chaindb(h)


def insertiondb_merge(paths, out_path):
    """`dentist merge-insertions` (commands/mergeInsertions.d:42-164): k-way merge of insertions.db files by
    (start contig, start part, end contig, end part); returns the number of insertions written."""
    L = lib()
    arr = (ctypes.c_char_p * max(1, len(paths)))(*[p.encode() for p in paths])
    n = ctypes.c_int64(0)
    L.dh_insertiondb_merge.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, ctypes.c_char_p,
                                       ctypes.POINTER(ctypes.c_int64)]
    _check(L.dh_insertiondb_merge(arr, len(paths), out_path.encode(), ctypes.byref(n)))
    return n.value
