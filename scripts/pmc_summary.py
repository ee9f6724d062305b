#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc counter-collection CSV (one counter per pass).

usage: pmc_summary.py <dir-with-csv> [<dir> ...]
Prints, per counter and kernel: launches, total (KB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE)
and the per-launch values in dispatch order.
"""
import csv
import glob
import os
import sys


def main(dirs):
    for d in dirs:
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            agg = {}
            with open(path, newline="") as f:
                rd = csv.DictReader(f)
                for row in rd:
                    name = row.get("Kernel_Name") or row.get("kernel_name") or "?"
                    name = name.split("(")[0]
                    cname = row.get("Counter_Name") or row.get("counter_name") or "?"
                    val = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                    did = int(row.get("Dispatch_Id") or row.get("dispatch_id") or 0)
                    agg.setdefault((cname, name), {}).setdefault(did, 0.0)
                    agg[(cname, name)][did] += val
            for (cname, name), per in sorted(agg.items()):
                vals = [per[k] for k in sorted(per)]
                tot = sum(vals)
                shown = ", ".join(str(int(v)) for v in vals[:24]) + (" ..." if len(vals) > 24 else "")
                print(f"{cname:11s} {name[:44]:44s} launches={len(vals):4d} total_KB={tot:16.0f}  per_launch_KB=[{shown}]")


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
