#!/bin/bash
# dev: kernel summary of ONE rank's mapping at N = 8 (125 000 reads of configs[2], scripts/dev/map_shard.py) under rocprofv3 --kernel-trace
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-shard_map}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sm
( cd "$root" && SHARD_MAP_MODES=default rocprofv3 --kernel-trace --stats -d /tmp/prof_sm -o run -- python scripts/dev/map_shard.py > "$out/map_shard.log" 2>&1 )
db=$(find /tmp/prof_sm -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python scripts/dev/map_shard.py (125 000 reads x 15 kb against the 100 Mb assembly, one chunk; 4 mappings)"; python "$root/scripts/rocpd_summary.py" "$db" | head -24; } > "$out/kernel_stats.txt"
cat "$out/kernel_stats.txt" | cut -c1-130
tail -2 "$out/map_shard.log"
