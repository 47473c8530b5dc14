#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA
utilisation (the north-star's "rocprof-reported MFMA utilisation for the GEMMs").

  MFMA busy share of the launch = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles), where kernel cycles =
  GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back) and the chip
  has 256 CUs x 4 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32).
  Also reported: the effective clock GRBM_GUI_ACTIVE / 8 / duration, and MFMA-busy cycles per second against
  1024 SIMDs x 2.4 GHz (= share of the 2.5 PFLOP/s dense bf16 peak the kernel's MFMA issue alone accounts for).

    python tools/pmc_mfma.py gpurun_out/pmc_mfma/*_counter_collection.csv profiles/r02_mfma_util.json
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:96]


def main():
    path, out = sys.argv[1:3]
    per = defaultdict(lambda: defaultdict(float))
    seen = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            did = r.get("Dispatch_Id")
            if did not in seen[k]:
                seen[k].add(did)
                per[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                per[k]["_n"] += 1
    rows = {}
    for k, c in per.items():
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        kern_cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        ns = c["_ns"]
        rows[k] = dict(launches=int(c["_n"]), avg_us_under_pmc=round(ns / c["_n"] / 1e3, 2),
                       mfma_busy_share=round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * kern_cyc), 4) if kern_cyc else None,
                       effective_clock_ghz=round(kern_cyc / ns, 3) if ns else None,
                       share_of_2p5_pflops_peak=round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * 2.4 * ns), 4) if ns else None)
    doc = dict(note="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py (one optimizer step); "
                    "mfma_busy_share = MFMA-busy cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8",
               kernels=dict(sorted(rows.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_us_under_pmc"])))
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in list(doc["kernels"].items())[:14]:
        print(f"{k[:64]:64s} n={v['launches']:5d} {v['avg_us_under_pmc']:9.1f} us  busy {v['mfma_busy_share']}  clk {v['effective_clock_ghz']} GHz  of-peak {v['share_of_2p5_pflops_peak']}")


if __name__ == "__main__":
    main()
