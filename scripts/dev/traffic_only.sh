#!/bin/bash
# dev: the HBM traffic passes of scripts/profile_round.sh alone (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) -> kernel_traffic.json
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-traffic}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -o run -- python "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --ref-steps 0 > "$out/bench_$c.log" 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), python bench.py --steps 1 --warmup 0 --no-cpu-baseline --ref-steps 0"; python "$root/scripts/pmc_summary.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE; } > "$out/pmc_hbm_traffic.txt"
python "$root/scripts/traffic_json.py" /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE "$out/kernel_traffic.json" cfg2_100Mb_1000gaps_1Mx15kb 8 20 1
