"""dentist_amd -- MI355X (gfx950) implementation of DENTIST's alignment + consensus hot path.

The product is ``libdentist_hip.so`` (hand-written HIP kernels behind the C ABI declared in
``include/dentist_hip.h``).  This package is the thin ctypes binding tests and ``bench.py`` use to
call that ABI; it contains no compute of its own and no CPU fallback -- loading fails loudly when
the library is missing, and every compute call fails with ``DhError`` when no HIP device is
usable.
"""
from ._lib import (AlignOpts, AlignStats, Context, DazzDb, Db, DhError, dazz_create_dam, dazz_create_db, dazz_split, dazz_write_mask, INSERTION_DTYPE, LA_DTYPE, Pileups, DeviceTrace,  # noqa: F401
                   ProcessOpts, default_align_opts, default_process_opts, las_read, las_write, lib, lib_path,
                   process_pileups, process_stats, output_fasta, output_assembly, tile_qv, consensus, las_merge, merge_las, Cropped, translate_trace_point, common_trace_point, pileupdb_write, pileupdb_read,
                   insertiondb_write, insertiondb_read, insertiondb_merge, collect_filter, scaffold_pileups, scaffold_spanning_pileups, scaffold_all_pileups, max_coverage_reads, max_improper_coverage_reads, propagate_mask, validate_regions, JOIN_DTYPE, READ_ALIGNMENT_DTYPE, SEEDED_DTYPE, CHAIN_LA_DTYPE, INSERTION_REC_DTYPE, Comm, shard_run)
