#!/usr/bin/env python3
"""Condenses the raw rocprofv3 output of tools/profile_round.sh into profiles/<tag>_*:
  <tag>_kernel_stats.csv   the --stats table restricted to this repo's kernels
  <tag>_hbm_traffic.json   per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
                           (gfx950 correction: FETCH_SIZE counts 64 B per 128 B request for wide
                           coalesced reads -> doubled; MI355X_MICROARCH.md section HBM)
  <tag>_sq_counters.csv    per-kernel averages of the SQ counters
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for k in ("k_stft", "k_istft", "k_encoder", "k_gtcn", "k_decoder", "k_state_convert"):
        if k in name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # --- kernel stats
    # gpurun merges every call's output into the same directory: only the newest file of a pass counts
    def newest(sub, pat):
        f = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
        return [max(f, key=os.path.getmtime)] if f else []
    f = newest("trace", "*kernel_stats.csv")
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if short(r["Name"])]
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
    # --- PMC
    def pmc(sub):
        acc = defaultdict(lambda: defaultdict(list))
        for f in newest(sub, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return acc
    fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
    # coalesced read (16 B per lane) -> doubled for the kernels whose reads are 16 B per lane
    # (k_decoder, k_gtcn); other access widths are "uncalibrated" there, so for the kernels that
    # read 4/8 B per lane the raw value is kept and the known byte count is quoted beside it.
    WIDE = {"k_decoder", "k_gtcn"}
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        fs = fetch.get(k, {}).get("FETCH_SIZE", [])
        ws = write.get(k, {}).get("WRITE_SIZE", [])
        fkb = sum(fs) / len(fs) if fs else None
        wkb = sum(ws) / len(ws) if ws else None
        mult = 2 if k in WIDE else 1
        traffic[k] = {
            "FETCH_SIZE_KB_raw_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb,
            "fetch_multiplier": mult,
            "hbm_bytes_per_launch": (mult * fkb * 1024 if fkb is not None else 0) + (wkb * 1024 if wkb is not None else 0),
            "note": ("FETCH_SIZE doubled: 16 B/lane coalesced reads are tallied at 64 B per 128 B request on gfx950"
                     if mult == 2 else "4/8 B per lane reads: FETCH_SIZE left raw (uncalibrated width)"),
            "launches_sampled": max(len(fs), len(ws)),
        }
    if traffic:
        # k_gtcn covers both stacks; bench.py looks up k_gtcn1/k_gtcn2 too
        if "k_gtcn" in traffic:
            traffic["k_gtcn1"] = traffic["k_gtcn2"] = traffic["k_gtcn"]
        json.dump(traffic, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
        print(json.dumps(traffic, indent=1))
    if sq:
        names = sorted({c for k in sq for c in sq[k]})
        with open(os.path.join(dst, f"{tag}_sq_counters.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["Kernel"] + names)
            for k in sorted(sq):
                w.writerow([k] + [f"{sum(sq[k][c]) / len(sq[k][c]):.0f}" if sq[k].get(c) else "" for c in names])
        print(open(os.path.join(dst, f"{tag}_sq_counters.csv")).read())


if __name__ == "__main__":
    main()
