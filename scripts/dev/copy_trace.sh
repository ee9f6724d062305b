#!/bin/bash
# dev: the memory copies of one bench step (direction, bytes, duration) next to the kernels they overlap
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ct
( cd "$root" && DH_PROCESS_SERIAL=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_ct -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --fast-steps 0 --ref-partners 0 > /tmp/ct.log 2>&1 )
db=$(find /tmp/prof_ct -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
mc = [t for t in tabs if "memory_cop" in t.lower() or "memcpy" in t.lower()]
print("tables:", mc[:6])
t = "memory_copies" if "memory_copies" in tabs else mc[0]
cols = [r[1] for r in cur.execute(f"pragma table_info({t})").fetchall()]
print(cols)
rows = cur.execute(f"select * from {t}").fetchall()
ks = cur.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(ks) if r[0].startswith("void k_mj_part") or r[0].startswith("k_mj_part")]
t0 = ks[idx[-2]][1] - 30e6 if len(idx) >= 2 else 0
ci = {c: i for i, c in enumerate(cols)}
s_i, e_i = ci.get("start"), ci.get("end")
out = []
for r in rows:
    if r[s_i] < t0: continue
    d = (r[e_i] - r[s_i]) / 1e6
    if d < 0.3: continue
    out.append((r[s_i], d, {c: r[ci[c]] for c in cols if c in ("name", "size", "src_agent_type", "dst_agent_type", "src_device", "dst_device", "kind")}))
for s, d, info in sorted(out):
    print("%9.2f ms  %7.2f ms  %s" % ((s - t0) / 1e6, d, info))
PY
