#!/usr/bin/env python3
"""Per-kernel averages of SQ instruction counters from rocprofv3 --pmc passes -> profiles/*_sampling_pmc_instructions.json.

usage: pmc_instructions.py <out.json> <note> <pass-dir> [<pass-dir> ...]

Each pass directory is what `rocprofv3 --pmc C1 C2 ... --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`
wrote.  A dispatch's rows (one per counter instance) are summed, then averaged per kernel name over the launches; only the
sampling kernels (k_rec_sample*, k_mut_sample*) are kept -- bench.py's `sampling_kernels` block sums their SQ_INSTS_VALU.
"""
import json
import sys
from collections import OrderedDict, defaultdict
import csv
import glob
import os

KEEP = ("k_rec_sample8", "k_mut_sample8", "k_rec_sample", "k_mut_sample")


def main():
    if len(sys.argv) < 4:
        raise SystemExit(__doc__)
    out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = OrderedDict()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            disp = defaultdict(float)
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    if name not in KEEP:
                        continue
                    disp[(name, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            agg = defaultdict(list)
            for (name, _, ctr), v in disp.items():
                agg[(name, ctr)].append(v)
            for (name, ctr), vs in agg.items():
                per.setdefault(name, OrderedDict())[ctr] = sum(vs) / len(vs)
                per[name]["launches_" + ctr] = len(vs)
    per["note"] = note
    with open(out, "w") as fh:
        json.dump(per, fh, indent=1)
    print(json.dumps({k: v.get("SQ_INSTS_VALU") for k, v in per.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
