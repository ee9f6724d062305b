#!/bin/bash
# kernel trace and FETCH_SIZE / WRITE_SIZE of one rank's share of configs[4] (scripts/cfg4_rank_share.py), for profiles/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_cfg4
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c4_kt -o run -- python "$root/scripts/cfg4_rank_share.py" 2 rebuild > "$out/line_kernel_trace.log" 2>&1
db=$(find /tmp/c4_kt -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python scripts/cfg4_rank_share.py 2 rebuild (3 passes in total, the first one sizes the buffers)"; python "$root/scripts/rocpd_summary.py" "$db"; } > "$out/kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/c4_$c -o run -- python "$root/scripts/cfg4_rank_share.py" 0 > "$out/line_$c.log" 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python scripts/cfg4_rank_share.py 0 (one pass)"; python "$root/scripts/pmc_summary.py" /tmp/c4_FETCH_SIZE /tmp/c4_WRITE_SIZE | grep -E "k_seed|k_tile|k_kmer|k_fat"; } > "$out/pmc_hbm_traffic.txt"
head -14 "$out/kernel_stats.txt"; cat "$out/pmc_hbm_traffic.txt" | cut -c1-200
