#!/bin/bash
# dev: kernel timeline of ONE emulated rank's process stage at N = 8 (scripts/dev/shard8.py under rocprofv3 --kernel-trace):
# every kernel of the window with the idle time in front of it on its stream, busy / wall at the end
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-tl8}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl8
( cd "$root" && rocprofv3 --kernel-trace -d /tmp/prof_tl8 -o run -- python scripts/dev/shard8.py > "$out/shard8.log" 2>&1 )
db=$(find /tmp/prof_tl8 -name "*.db" | head -1)
python - "$db" > "$out/timeline.txt" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = cur.execute(f"select name, start, end, {sid or 0}, grid_x, workgroup_x from kernels order by start").fetchall()
# one k_pile_funnel per process call: the window of emulated rank 3 = from the last kernel before its pile DB's k_join_part ...
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_pile_funnel")]
print("# process calls seen:", len(idx))
if len(idx) >= 5:
    # the window: from the end of call 2's last kernel to the end of call 3's last kernel (k_pile_funnel sits early in a call)
    def call_end(k):
        nxt = idx[k + 1] if k + 1 < len(idx) else len(rows)
        # the next call starts with its pile DB kernels: find the largest idle gap between funnel k and funnel k+1
        best, at = -1, idx[k]
        for i in range(idx[k] + 1, nxt):
            g = rows[i][1] - max(r[2] for r in rows[max(idx[k], i - 8):i])
            if g > best:
                best, at = g, i
        return at
    i0, i1 = call_end(2), call_end(3)
    t0 = rows[i0][1]
    streams = sorted({r[3] for r in rows[i0:i1]})
    last_end = {}
    busy = 0
    cover_end = t0
    for name, s, e, q, gx, wx in rows[i0:i1]:
        d = (e - s) / 1e6
        col = streams.index(q)
        print(f"{(s - t0) / 1e6:9.3f} ms  q{col}  dur {d:8.3f}  gap_on_q {((s - last_end.get(q, s)) / 1e6):8.3f}  {name.split('(')[0][:48]}  grid {gx} wg {wx}")
        last_end[q] = e
        if e > cover_end:
            busy += e - max(s, cover_end)
            cover_end = e
    print(f"# window {(rows[i1][1] - t0) / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms, {i1 - i0} kernels")
PY
tail -3 "$out/timeline.txt"
