#!/bin/bash
# dev: kernel timeline of the last bench step by stream (two concurrent halves of the process stage)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-tl}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
( cd "$root" && rocprofv3 --kernel-trace -d /tmp/prof_tl -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fast-steps 0 --ref-partners 0 > "$out/bench.log" 2>&1 )
db=$(find /tmp/prof_tl -name "*.db" | head -1)
python - "$db" > "$out/timeline.txt" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print("# columns:", cols)
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = cur.execute(f"select name, start, end, {sid or 0}, grid_x, workgroup_x from kernels order by start").fetchall()
# the last step: from the last k_kmer_pass<true> preceded by a long mapping... simply the last 45 % of the trace
idx = [i for i, r in enumerate(rows) if "k_mj_part" in r[0]] or [i for i, r in enumerate(rows) if r[0].startswith("void k_seed<1024") and r[2] - r[1] > 10e6]
i0 = idx[-2] if len(idx) >= 2 else 0   # (two mapping chunks per step: the first one of the last step)
t0 = rows[i0][1]
streams = sorted({r[3] for r in rows[i0:]})
print("# streams:", streams)
last_end = {}
for name, s, e, q, gx, wx in rows[i0:]:
    d = (e - s) / 1e6
    if d < 0.2:
        continue
    col = streams.index(q)
    print(f"{(s - t0) / 1e6:9.3f} ms  q{col}  dur {d:8.3f}  gap_on_q {((s - last_end.get(q, s)) / 1e6):8.3f}  {name.split('(')[0][:48]}  grid {gx} wg {wx}")
    last_end[q] = e
# windows in which no kernel ran on any stream, longest first
ivs = sorted((s, e) for name, s, e, q, gx, wx in rows[i0:])
idle, cover = [], ivs[0][1]
for s, e in ivs[1:]:
    if s > cover:
        idle.append((s - cover, cover))
    cover = max(cover, e)
print("# step window %.1f ms, device idle %.1f ms; longest idle windows (ms, at):" % ((cover - t0) / 1e6, sum(d for d, _ in idle) / 1e6),
      [(round(d / 1e6, 2), round((a - t0) / 1e6, 1)) for d, a in sorted(idle, reverse=True)[:14]])
PY
tail -1 "$out/bench.log" | cut -c1-200
