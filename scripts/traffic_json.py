#!/usr/bin/env python3
"""HBM traffic per launch of the seed and extension kernels of one bench step, from the two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass) of scripts/profile_round.sh.

usage: traffic_json.py <fetch-dir> <write-dir> <out.json> workload kmer_mod k algo

Written as profiles/<round>_kernel_traffic.json, which bench.py reads for `roofline.traffic` when its
configuration matches.  The counters are taken as rocprofv3 reports them (KB); the x2 correction of the
microarchitecture guide applies to wide coalesced streaming reads only: it is applied to k_mj_part's
FETCH_SIZE (the one kernel here that streams -- the read bytes) and to nothing else: the seed kernels read
8 / 16 bytes at scattered lines, k_tile 4/8-byte words per lane.  Mapping launches = the dispatches
before the first crop kernel (k_gather_parts) of the step.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_build_id():
    """sha1 over the kernel sources -- the .hip files and the headers they include (dh_device.h, dh_join.h, dh_kmer.h,
    dh_mjoin.h, dh_tile.h); not the host-only headers dh_internal.h / dh_parallel.h -- : bench.py compares it with the tree it
    runs from, so that a traffic figure taken on another build of the kernels is never reported as this build's."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "dentist_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) and name not in ("dh_internal.h", "dh_parallel.h"):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def rows(d):
    out = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = (row.get("Kernel_Name") or row.get("kernel_name") or "?").split("(")[0]
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0)
                did = int(row.get("Dispatch_Id") or row.get("dispatch_id") or 0)
                out.append((did, name, val))
    return out


def per_kernel(d, fetch=False):
    per = {}
    for did, name, val in rows(d):
        per.setdefault(did, [name, 0.0])[1] += val
    order = sorted(per)
    first_crop = next((i for i in order if per[i][0].strip() == "k_gather_parts"), None)
    res = {}
    for i in order:
        name, kb = per[i]
        # the seeds of a mapping chunk: the directory lookups (k_seed<cap, false>) or, since round 5, the partitioned join
        # (k_mj_*) with the segment-fed back end (k_seed<cap, true>) -- one family, counted per chunk
        # (templated kernels are named `void k_mj_part<1, false>(...)`: matched by substring)
        fam = ("k_seed" if ("k_seed<" in name and "k_seed<0" not in name) or "k_mj_" in name else
               ("k_tile" if name.strip() == "k_tile" or "k_tile<" in name else ("k_seed_redo" if "k_seed<0" in name else None)))
        if "k_mj_part" in name:
            res.setdefault(("chunks", "mapping"), [0, 0.0])[0] += 1
        if fam is None:
            continue
        stage = "mapping" if first_crop is None or i < first_crop else "process"
        r = res.setdefault((fam, stage), [0, 0.0])
        r[0] += 1
        # FETCH_SIZE of k_mj_part: it streams the chunk's bases with 8-byte loads per lane, 512-byte wavefront requests --
        # the counter reports 4.3 GB for a launch that reads 7.87 GB of bases (profiles/r06a_ref_pmc_hbm_traffic.txt), the
        # half-counting of wide coalesced reads the microarchitecture guide describes: x2.  Every other kernel of the two
        # families loads 8 / 16 bytes at scattered lines (calibrated 0.996, profiles/r04b_fetch_size_calibration.txt)
        r[1] += kb * 1024.0 * (2.0 if fetch and "k_mj_part" in name else 1.0)
    return res


def main(fetch_dir, write_dir, out, workload, kmer_mod, k, algo):
    f, w = per_kernel(fetch_dir, fetch=True), per_kernel(write_dir)
    j = {"workload": workload, "mapping_kmer_mod": int(kmer_mod), "mapping_k": int(k), "mapping_algo": int(algo),
         "kernel_build_id": kernel_build_id(),
         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE of k_mj_part x2 (wide coalesced "
                   "stream of the read bytes, half-counted), everything else as reported (8 / 16-byte loads at scattered lines)",
         "launches": {}}
    for key in sorted(set(f) | set(w)):
        n = (f.get(key) or w.get(key))[0]
        fb, wb = (f.get(key) or [0, 0.0])[1], (w.get(key) or [0, 0.0])[1]
        j["launches"]["%s/%s" % key] = {"launches": n, "fetch_bytes": fb, "write_bytes": wb}
    def per_launch(fam):
        n = sum(v["launches"] for k_, v in j["launches"].items() if k_.startswith(fam + "/mapping"))
        if fam == "k_seed" and "chunks/mapping" in j["launches"]:
            n = j["launches"]["chunks/mapping"]["launches"]   # the join's kernels of one chunk count as one launch
        b = sum(v["fetch_bytes"] + v["write_bytes"] for k_, v in j["launches"].items()
                if k_.startswith(fam + "/mapping") or k_.startswith(fam + "_redo/mapping"))
        return b / n if n else None
    j["k_seed_hbm_bytes_per_launch"] = per_launch("k_seed")
    nt = sum(v["launches"] for k_, v in j["launches"].items() if k_.startswith("k_tile/"))
    bt = sum(v["fetch_bytes"] + v["write_bytes"] for k_, v in j["launches"].items() if k_.startswith("k_tile/"))
    j["k_tile_hbm_bytes_per_launch"] = bt / nt if nt else None      # all k_tile launches of the step, like roofline_tile
    with open(out, "w") as fo:
        json.dump(j, fo, indent=1)
        fo.write("\n")
    print(json.dumps({k_: j[k_] for k_ in ("k_seed_hbm_bytes_per_launch", "k_tile_hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main(*sys.argv[1:8])
