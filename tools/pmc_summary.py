#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/*_pmc_hbm.json.

usage: pmc_summary.py <dir-with-FETCH_SIZE-pass> <dir-with-WRITE_SIZE-pass> <out.json> <algorithmic-bytes-per-stitch-launch> [note]

Each pass directory is what `rocprofv3 --pmc X --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`
wrote (one *_counter_collection.csv below it).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per
dispatch; a dispatch's rows (one per counter instance) are summed first, then averaged per kernel name.
The stitch summary applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts
half of the bytes of a wide (16 B/lane) streaming read, WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict, defaultdict


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    per_dispatch = OrderedDict()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                key = (f, row["Dispatch_Id"])
                name = row["Kernel_Name"].split("(")[0]
                if key not in per_dispatch:
                    per_dispatch[key] = [name, 0.0]
                per_dispatch[key][1] += float(row["Counter_Value"])
    agg = defaultdict(list)
    order = []
    for name, v in per_dispatch.values():
        if name not in agg:
            order.append(name)
        agg[name].append(v)
    return OrderedDict((n, {"dispatches": len(agg[n]), "avg_KiB": sum(agg[n]) / len(agg[n])}) for n in order)


def main():
    if len(sys.argv) < 5:
        raise SystemExit(__doc__)
    fetch_dir, write_dir, out, alg = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    res = OrderedDict()
    res["FETCH_SIZE"] = per_kernel(fetch_dir, "FETCH_SIZE")
    res["WRITE_SIZE"] = per_kernel(write_dir, "WRITE_SIZE")
    stitch = ([k for k in res["FETCH_SIZE"] if "k_stitch_segments" in k] or [k for k in res["FETCH_SIZE"] if "k_stitch_regions" in k]
              or [k for k in res["FETCH_SIZE"] if "k_stitch_parent" in k])
    if stitch:
        k = stitch[0]
        fetch = res["FETCH_SIZE"][k]["avg_KiB"] * 1024.0
        write = res["WRITE_SIZE"][k]["avg_KiB"] * 1024.0
        traffic = 2.0 * fetch + write
        res["stitch_summary"] = {
            "kernel": k,
            "fetch_bytes_raw": fetch,
            "fetch_bytes_corrected_x2_gfx950_wide_stream": 2.0 * fetch,
            "write_bytes": write,
            "hbm_traffic_bytes_per_launch": traffic,
            "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": traffic / alg,
            "note": note,
        }
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res.get("stitch_summary", {}), indent=1))


if __name__ == "__main__":
    main()
