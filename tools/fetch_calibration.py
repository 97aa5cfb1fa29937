#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration from tools/ubench_fetch_size.hip run under rocprofv3 (two passes):
    python tools/fetch_calibration.py gpurun_out/<tag>/fetch_cal_f gpurun_out/<tag>/fetch_cal_w > profiles/<tag>_fetch_calibration.json
Every kernel of the microbenchmark moves exactly 1 GiB per direction per launch; the factor of a width is
true bytes / (counter value in KB x 1024)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def label(name):
    def width(n):
        m = re.search(r"__vector\((\d)\)|ext_vector_type\((\d)\)", n)
        return 4 * int(m.group(1) or m.group(2)) if m else 4
    if "rd_bf16x4" in name:
        return "read 8 B/lane (4 x bf16 decoded, nontemporal)"
    if "rd<" in name:
        head = name[name.index("rd<"):name.index(">(") + 1]
        return f"read {width(head)} B/lane ({'nontemporal' if 'true>' in head else 'plain'})"
    if "cp<" in name:
        head = name[name.index("cp<"):name.index(">(") + 1]
        return f"copy {width(head)} B/lane"
    return None


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    gib = float(1 << 30)
    out = {"note": "tools/ubench_fetch_size.hip: every launch reads (and, copy kernels, writes) exactly 1 GiB, four times "
                   "the Infinity Cache; factor = true bytes / (counter KB x 1024); launches: one warm-up + one timed per kernel"}
    for d, counter, key in ((fdir, "FETCH_SIZE", "fetch"), (wdir, "WRITE_SIZE", "write")):
        for name, vals in collect(d, counter).items():
            lab = label(name)
            if not lab or (key == "write" and not lab.startswith("copy")):
                continue
            kb = sum(vals) / len(vals)
            e = out.setdefault(lab, {})
            e[f"{counter}_KB_per_launch"] = round(kb, 1)
            e[f"{key}_factor"] = round(gib / (kb * 1024.0), 4) if kb else None
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
