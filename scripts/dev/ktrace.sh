# dev: kernel-trace summary of the bench (pass env such as DH_PROCESS_SERIAL=1 in front): bash scripts/dev/ktrace.sh <tag> [bench args]
tag=${1:-kt}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o run -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$root/gpurun_out/${tag}_bench.log" 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python "$root/scripts/rocpd_summary.py" "$db" > "$root/gpurun_out/${tag}_kernel_stats.txt"
tail -1 "$root/gpurun_out/${tag}_bench.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])"
head -45 "$root/gpurun_out/${tag}_kernel_stats.txt"
