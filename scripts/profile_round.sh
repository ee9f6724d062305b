#!/bin/bash
# Collects the profiles committed under profiles/: kernel trace summary, HBM traffic counters and
# the SQ instruction / occupancy counters of the wave and seed kernels (separate --pmc passes, never
# combined with other trace domains).  Run on the GPU box:
#   bash scripts/profile_round.sh <tag> [bench args...]
set -u
tag=${1:-r02}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o run -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/bench_kernel_trace.log" 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline $* (4 steps in total)"; python "$root/scripts/rocpd_summary.py" "$db"; } > "$out/kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o run -- python "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/bench_$c.log" 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), python bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"; python "$root/scripts/pmc_summary.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE; } > "$out/pmc_hbm_traffic.txt"
python "$root/scripts/traffic_json.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE "$out/kernel_traffic.json" "${DH_PROF_WORKLOAD:-cfg2_100Mb_1000gaps_1Mx15kb}" "${DH_PROF_KMER_MOD:-1}" "${DH_PROF_K:-20}" "${DH_PROF_ALGO:-1}"
# SQ counters (8 slots per pass): instruction mix / issue, then waits and LDS conflicts
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_SQ1 -o run -- python "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/bench_SQ1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d /tmp/prof_SQ2 -o run -- python "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/bench_SQ2.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <SQ counters> (two separate passes of 8), python bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"; python "$root/scripts/pmc_summary.py" /tmp/prof_SQ1 /tmp/prof_SQ2 | grep -E "k_tile|k_wave|k_seed|k_mj_|k_seg_vote|k_compact|k_join|k_pile"; } > "$out/pmc_sq_counters.txt"
tail -1 "$out/bench_kernel_trace.log" | cut -c1-300
head -30 "$out/kernel_stats.txt"
grep -E "k_tile|k_wave|k_seed|k_mj_" "$out/pmc_hbm_traffic.txt"
cat "$out/pmc_sq_counters.txt" | cut -c1-200
