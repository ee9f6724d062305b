#!/bin/bash
# dev: the process stage of configs[2] alone (serial, one context): lap times of the host thread (DH_TRACE) and the
# kernel list of the same run in launch order with gaps (where the device waits for the host)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-proc}
mkdir -p "$out"
cd "$root"
DH_TRACE=1 python scripts/dev/pile_only.py 3 > "$out/laps.log" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_po
( cd "$root" && rocprofv3 --kernel-trace --stats -d /tmp/prof_po -o run -- python scripts/dev/pile_only.py 2 > "$out/kt.log" 2>&1 )
db=$(find /tmp/prof_po -name "*.db" | head -1)
python "$root/scripts/rocpd_summary.py" "$db" > "$out/kernel_stats.txt"
python - "$db" > "$out/timeline.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# the last process call: find the last k_gather_parts (crop) and print from there
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_gather_parts")]
i0 = idx[-1] if idx else 0
t0 = rows[i0][1]
prev_end = t0
for name, s, e in rows[i0:]:
    print(f"{(s - t0) / 1e6:9.3f} ms  +gap {(s - prev_end) / 1e6:8.3f}  dur {(e - s) / 1e6:8.3f}  {name.split('(')[0][:50]}")
    prev_end = max(prev_end, e)
PY
tail -5 "$out/laps.log"
