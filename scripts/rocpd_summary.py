#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd SQLite result."""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name.split("(")[0], [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:60]:60s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:11.2f} {a[2] / 1e3:11.2f} "
              f"{a[3] / 1e3:11.2f} {100.0 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
