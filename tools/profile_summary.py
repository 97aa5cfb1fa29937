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
    for k in ("k_stream_wide", "k_stream_ms", "k_front", "k_stft", "k_istft", "k_encoder", "k_gtcn_ms", "k_gtcn", "k_decoder", "k_state_convert"):
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
    for k, v in pmc("pmc_sq2").items():                 # second SQ pass: same kernels, other counters
        sq[k].update(v)
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
    # coalesced read (16 B per lane) -> doubled for the kernels whose reads are 16 B per lane
    # (k_decoder, k_gtcn); other access widths are "uncalibrated" there, so for the kernels that
    # read 4/8 B per lane the raw value is kept and the known byte count is quoted beside it.
    # Round 5 calibrated the other widths (tools/ubench_fetch_size.hip -> profiles/r05_fetch_calibration.json): a fully
    # coalesced streaming read is tallied at HALF its bytes at 4, 8 and 16 bytes per lane alike, plain or nontemporal
    # (factor 2.000 +- 1e-4 on 1 GiB); WRITE_SIZE is exact at every width.  One multiplier for every kernel.
    WIDE = {"k_decoder", "k_gtcn", "k_encoder", "k_front", "k_istft", "k_stft", "k_stream_ms", "k_stream_wide", "k_gtcn_ms", "k_state_convert"}
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
            "note": ("FETCH_SIZE doubled: coalesced streaming reads are tallied at half their bytes on gfx950 at 4, 8 and 16 B "
                     "per lane (calibrated: profiles/r05_fetch_calibration.json)"
                     if mult == 2 else "FETCH_SIZE left raw"),
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
    # --- train step: HBM counters per kernel next to its average time (tools/profile_round.sh train_pmc_* passes)
    import re

    def tshort(name):       # any kernel of this repo, with its template arguments
        mm = re.search(r"(k_\w+(?:<[^>(]*>)?)", name)
        return mm.group(1) if mm else None

    def tpmc(sub):
        acc = defaultdict(lambda: defaultdict(list))
        for f in newest(sub, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = tshort(r["Kernel_Name"])
                if k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return acc
    tf, tw = tpmc("train_pmc_fetch"), tpmc("train_pmc_write")
    if tf or tw:
        f = newest("train_f32", "*kernel_stats.csv")
        rows = list(csv.DictReader(open(f[0]))) if f else []
        t_ns = {tshort(r["Name"]): float(r["AverageNs"]) for r in rows if tshort(r["Name"])}
        calls = {tshort(r["Name"]): int(r["Calls"]) for r in rows if tshort(r["Name"])}
        out = {"note": "FETCH_SIZE / WRITE_SIZE in KB per launch, averaged over the launches of a kernel (layers of "
                       "different sizes share a kernel); FETCH_SIZE is tallied at half the bytes for 16 B-per-lane "
                       "coalesced reads on gfx950 (MI355X_MICROARCH.md), so read_bytes_x2 is the figure to use for the "
                       "kernels that load 16 bytes per lane (all the large ones here); TB_per_s = (2 x fetch + write) / "
                       "average time"}
        tot = 0.0
        for k in sorted(set(tf) | set(tw), key=lambda k: -calls.get(k, 0) * t_ns.get(k, 0.0)):
            fs = tf.get(k, {}).get("FETCH_SIZE", [])
            ws = tw.get(k, {}).get("WRITE_SIZE", [])
            if k not in t_ns or not (fs or ws):
                continue
            fkb = sum(fs) / len(fs) if fs else 0.0
            wkb = sum(ws) / len(ws) if ws else 0.0
            byt = (2 * fkb + wkb) * 1024
            out[k] = {"launches_per_profile": calls[k], "avg_us": round(t_ns[k] / 1e3, 1),
                      "FETCH_SIZE_KB_raw": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1),
                      "read_bytes_x2_plus_write_MB": round(byt / 1e6, 1),
                      "TB_per_s": round(byt / t_ns[k] / 1e3, 2)}
            tot += byt * calls[k]
        steps = 5       # bench.py --mode train --steps 3 --warmup 1 runs 1 + 1 + 3 steps
        out["total_GB_per_step"] = round(tot / steps / 1e9, 1)
        # Every storage mode, dispatch by dispatch (no kernel averages): the counter passes run --steps 2 --warmup 1 after
        # one forward + backward of validation = 4 step equivalents.  FETCH_SIZE x 2 is calibrated for 16-byte-per-lane
        # reads only (MI355X_MICROARCH.md); the 16-bit tensors are read 8 bytes per lane, for which the guide gives no
        # factor -- both readings are reported, the truth for the 16-bit modes lies between them.
        modes = {}
        for mode, fsub, wsub in (("f32", "train_pmc_fetch", "train_pmc_write"), ("bf16", "train_bf16_pmc_fetch", "train_bf16_pmc_write"),
                                 ("bf16_saves", "train_bf16_saves_pmc_fetch", "train_bf16_saves_pmc_write"),
                                 ("bf16_grads", "train_bf16_grads_pmc_fetch", "train_bf16_grads_pmc_write")):
            tot_f = tot_w = 0.0
            for sub, name in ((fsub, "FETCH_SIZE"), (wsub, "WRITE_SIZE")):
                for f in newest(sub, "*counter_collection.csv"):
                    for r in csv.DictReader(open(f)):
                        if r["Counter_Name"] == name and tshort(r["Kernel_Name"]):
                            if name == "FETCH_SIZE": tot_f += float(r["Counter_Value"])
                            else: tot_w += float(r["Counter_Value"])
            if tot_f or tot_w:
                modes[mode] = {"FETCH_SIZE_GB_raw_per_step": round(tot_f * 1024 / 4 / 1e9, 2),
                               "WRITE_SIZE_GB_per_step": round(tot_w * 1024 / 4 / 1e9, 2),
                               # ONE figure since round 5: the x2 holds for the 8-byte-per-lane reads of the 16-bit
                               # tensors too (profiles/r05_fetch_calibration.json)
                               "GB_per_step": round((2 * tot_f + tot_w) * 1024 / 4 / 1e9, 1)}
        out["per_storage_mode"] = modes
        json.dump(out, open(os.path.join(dst, f"{tag}_train_hbm_traffic.json"), "w"), indent=1)
        print("train traffic: %.1f GB per step" % out["total_GB_per_step"], json.dumps(modes))
    # --- the single-launch streaming step under the HBM counters at several stream counts (state inside / past the
    # Infinity Cache): bytes per frame-step against SURVEY 8d's 94 KB state-traffic figure
    sres = {}
    for N in (1024, 16384, 65536):
        fs = [v for k, c in pmc(f"stream_pmc_{N}_fetch").items() if k in ("k_stream_ms", "k_stream_wide") for v in c.get("FETCH_SIZE", [])]
        ws = [v for k, c in pmc(f"stream_pmc_{N}_write").items() if k in ("k_stream_ms", "k_stream_wide") for v in c.get("WRITE_SIZE", [])]
        if not fs and not ws:
            continue
        fkb = sum(fs) / len(fs) if fs else 0.0
        wkb = sum(ws) / len(ws) if ws else 0.0
        byt = (2 * fkb + wkb) * 1024                           # 16-byte-per-lane reads: FETCH_SIZE x 2
        sres[str(N)] = {"launches_sampled": max(len(fs), len(ws)), "FETCH_SIZE_KB_raw_per_step": round(fkb, 1),
                        "WRITE_SIZE_KB_per_step": round(wkb, 1), "hbm_bytes_per_step": round(byt),
                        "bytes_per_frame_step": round(byt / N, 1), "of_94KB_state_figure": round(byt / N / (94 * 1024), 3),
                        "state_MiB": round(N * 152464 / 2 ** 20, 1)}
    if sres:
        sres["note"] = ("k_stream_ms (N <= 4 096 here) / k_stream_wide (seven streams per workgroup, picked by the library at N = 16 384 and 65 536), one launch per single-frame step of N streams; FETCH_SIZE x 2 + WRITE_SIZE per launch "
                        "(the counters sit on the L2's memory side: Infinity-Cache hits are counted, MI355X_MICROARCH.md)")
        json.dump(sres, open(os.path.join(dst, f"{tag}_stream_hbm_traffic.json"), "w"), indent=1)
        print(json.dumps(sres, indent=1))
    # --- derived figures per kernel: executed matrix FLOP, pipe busy share, co-execution share, effective clock
    f = newest("trace", "*kernel_stats.csv")
    avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(f[0])) if short(r["Name"])} if f else {}
    derived = {}
    for k in sorted(sq):
        c = {n: (sum(v) / len(v) if v else None) for n, v in sq[k].items()}
        d = {}
        if c.get("SQ_INSTS_MFMA") and avg_ns.get(k):
            if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
                # mixed matrix instructions (round 3: the decoder's dense 3x3 and the fused streaming step issue
                # v_mfma_f32_16x16x32_bf16, 16 384 FLOP, next to v_mfma_f32_16x16x4_f32, 2 048 FLOP): SQ_INSTS_MFMA
                # counts both kinds, the MOPS counters (512 FLOP each) separate them
                d["mfma_mops_f32"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32")
                d["mfma_mops_bf16"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"]
                d["mfma_flop_executed_per_launch"] = 512.0 * ((c.get("SQ_INSTS_VALU_MFMA_MOPS_F32") or 0) + c["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
            else:
                d["mfma_flop_executed_per_launch"] = c["SQ_INSTS_MFMA"] * 2048.0
            d["mfma_tflops_executed"] = d["mfma_flop_executed_per_launch"] / (avg_ns[k] * 1e-9) / 1e12
        if c.get("SQ_INSTS_VALU") and c.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_BUSY_CU_CYCLES"):
            # BUSY_CU_CYCLES counts per CU, MFMA_BUSY per SIMD-quad group: report the raw ratio and both terms
            d["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            d["busy_cu_cycles"] = c["SQ_BUSY_CU_CYCLES"]
            d["mfma_busy_over_busy_cu"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CU_CYCLES"]
        if c.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["coexec_over_mfma_busy"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
        if c.get("GRBM_GUI_ACTIVE") and avg_ns.get(k):
            d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns[k]
            d["effective_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / kernel time; reads high below ~0.3 ms (MI355X_MICROARCH.md)"
        if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES"):
            d["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if d:
            d["avg_launch_us"] = avg_ns.get(k, 0) / 1e3
            derived[k] = d
    if derived:
        json.dump(derived, open(os.path.join(dst, f"{tag}_derived.json"), "w"), indent=1)
        print(json.dumps(derived, indent=1))
    # --- the issue microbenchmark under PMC (tools/ubench_mfma_valu.hip): one row per launch, in launch order
    ub = newest("ub_pmc", "*counter_collection.csv")
    if ub:
        rows = defaultdict(dict)
        for r in csv.DictReader(open(ub[0])):
            rows[(int(r["Dispatch_Id"]), r["Kernel_Name"][:48])][r["Counter_Name"]] = float(r["Counter_Value"])
        with open(os.path.join(dst, f"{tag}_ubench_mfma_valu_pmc.csv"), "w", newline="") as o:
            names = sorted({c for v in rows.values() for c in v})
            w = csv.writer(o)
            w.writerow(["Dispatch", "Kernel"] + names)
            for (d, kn), v in sorted(rows.items()):
                w.writerow([d, kn] + [f"{v.get(c, 0):.0f}" for c in names])
        plain = os.path.join(src, "ub_plain.log")
        if os.path.exists(plain):
            open(os.path.join(dst, f"{tag}_ubench_mfma_valu.txt"), "w").write(open(plain).read())
    # --- streaming / training traces
    for sub, pats in (("stream", ("k_stream_ms", "k_stream_wide", "k_encoder", "k_gtcn_ms", "k_decoder")),
                      ("stream_wide", ("k_stream_ms", "k_stream_wide")), ("train_f32", None), ("train_bf16", None),
                      ("train_bf16_saves", None), ("train_bf16_grads", None)):
        f = newest(sub, "*kernel_stats.csv")
        if not f:
            continue
        rows = list(csv.DictReader(open(f[0])))
        with open(os.path.join(dst, f"{tag}_{sub}_kernel_stats.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for r in rows[:40]:
                import re
                mm = re.search(r"(k_\w+(?:<[^>(]*>)?)", r["Name"])
                nm = mm.group(1) if mm else r["Name"][:60]
                if pats is None or any(p_ in nm for p_ in pats):
                    w.writerow([nm, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])


if __name__ == "__main__":
    main()
