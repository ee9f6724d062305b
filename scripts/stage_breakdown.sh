#!/bin/bash
# Kernel-level breakdown of ONE bench step with the process stage's parts run one after the other (DH_PROCESS_SERIAL=1:
# no two kernels share the device, so a kernel's duration is its own) -- the stage-by-stage table of the pile-up
# all-vs-all at the reference's behaviour (n = 166 reads per pile-up).  Run on the GPU box:
#   bash scripts/stage_breakdown.sh <tag> [bench args...]
set -u
tag=${1:-r06}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/stage_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sb
( cd "$root" && DH_PROCESS_SERIAL=1 DH_TRACE=1 rocprofv3 --kernel-trace -d /tmp/prof_sb -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --fast-steps 0 --ref-partners 0 "$@" > "$out/bench.log" 2> "$out/bench.err" )
db=$(find /tmp/prof_sb -name "*.db" | head -1)
python - "$db" "$*" > "$out/stage_breakdown.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in rows]
# the last step: from its first k_kmer_pass / k_mj_tile_reads of the mapping (two mapping chunks per step)
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_mj_part")]
i0 = idx[-2] if len(idx) >= 2 else 0
while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 2e6:   # back over the index build in front of the first chunk
    i0 -= 1
step = rows[i0:]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"# DH_PROCESS_SERIAL=1 rocprofv3 --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --fast-steps 0 --ref-partners 0 {sys.argv[2]}")
print(f"# last step: {len(step)} kernels, window {(t1 - t0) / 1e6:.1f} ms, kernel time {sum(e - s for _, s, e in step) / 1e6:.1f} ms")
# phases: mapping = up to the last k_tile that follows a k_mj_* kernel; the rest is the process stage
last_mj = max(i for i, r in enumerate(step) if r[0].startswith("k_mj_") or r[0].startswith("k_seed<512, true") or r[0].startswith("k_seed<2048, true"))
map_end = next(i for i in range(last_mj, len(step)) if step[i][0].startswith("k_tile"))
for title, part in (("mapping", step[:map_end + 1]), ("process stage (parts one after the other)", step[map_end + 1:])):
    agg = {}
    for n, s, e in part:
        a = agg.setdefault(n, [0, 0, 0])
        a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s)
    tot = sum(a[1] for a in agg.values()) or 1
    w0, w1 = part[0][1], max(r[2] for r in part)
    print(f"\n## {title}: window {(w1 - w0) / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms")
    print(f"{'kernel':44s} {'calls':>6s} {'total_ms':>10s} {'max_ms':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / 1e6 < 0.3: continue
        print(f"{n[:44]:44s} {a[0]:6d} {a[1] / 1e6:10.2f} {a[2] / 1e6:9.2f} {100.0 * a[1] / tot:6.2f}")
print("\n## mapping, kernels >= 1 ms in launch order (ms from the step's first kernel)")
for n, s, e in step[:map_end + 1]:
    if e - s >= 1e6:
        print(f"{(s - step[0][1]) / 1e6:9.2f}  {(e - s) / 1e6:8.2f}  {n[:60]}")
# timeline of the process stage: kernels of at least 1 ms in launch order (the first pile-up batch shows the sequence)
print("\n## process stage, kernels >= 1 ms in launch order (ms from the stage's first kernel)")
p = step[map_end + 1:]
for n, s, e in p:
    if e - s >= 1e6:
        print(f"{(s - p[0][1]) / 1e6:9.2f}  {(e - s) / 1e6:8.2f}  {n[:60]}")
PY
tail -1 "$out/bench.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'stages_ms': d['stages_ms']}))" >> "$out/stage_breakdown.txt"
grep -E "^\[" "$out/bench.err" | tail -60 > "$out/trace_tail.txt"
head -120 "$out/stage_breakdown.txt"
