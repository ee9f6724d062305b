#!/bin/bash
# dev: SQ counters of the process stage alone (scripts/dev/pile_only.py, serial) + FETCH_SIZE calibration of random 16-byte loads
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-pmc_pile}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/rand_probe "$root/scripts/rand_access_probe.cpp"
{
  echo "# /tmp/rand_probe 16 (plain run)"; /tmp/rand_probe 16
  for b in 8 16; do
    rm -rf /tmp/rp_$b
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rp_$b -o run -- /tmp/rand_probe $b 1024 > /tmp/rp_$b.log 2>&1
    echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- rand_probe $b 1024   (two launches of 2^31 loads each)"
    grep "working set" /tmp/rp_$b.log
    python "$root/scripts/pmc_summary.py" /tmp/rp_$b | cut -c1-220
  done
} > "$out/fetch_size_calibration.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pp_$i
  ( cd "$root" && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pp_$i -o run -- python scripts/dev/pile_only.py 1 > /tmp/pp_$i.log 2>&1 )
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ counters> (two passes of 8) -- python scripts/dev/pile_only.py 1  (mapping of configs[2] + ONE serial process call; the k_tile / k_seed rows mix mapping and pile-up launches: per-launch values in dispatch order)"; python "$root/scripts/pmc_summary.py" /tmp/pp_1 /tmp/pp_2 | grep -E "k_tile|k_seed|k_join|k_seg_vote|k_compact_sym|k_dust" | cut -c1-400; } > "$out/pmc_sq_process.txt"
tail -3 "$out/fetch_size_calibration.txt" | cut -c1-200
