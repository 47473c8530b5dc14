#!/usr/bin/env python3
"""Launch sequence of one bench run from a rocprofv3 kernel trace: per launch the kernel (short name), grid, duration and the gap
to the previous launch.   python tools/trace_seq.py <kernel_trace.csv> [--match tn_bf16] [--context 1] [--limit 400]"""
import argparse
import csv
import re


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "")
    return n[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--match", default="")
    ap.add_argument("--context", type=int, default=0)
    ap.add_argument("--limit", type=int, default=400)
    ap.add_argument("--skip", type=int, default=0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = set()
    for i, r in enumerate(rows):
        if not a.match or a.match in r["Kernel_Name"]:
            for j in range(max(0, i - a.context), min(len(rows), i + a.context + 1)):
                keep.add(j)
    n = 0
    prev_end = None
    for i, r in enumerate(rows):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if i in keep:
            n += 1
            if n > a.skip:
                gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
                grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
                print(f"{i:6d} {short(r['Kernel_Name']):70s} grid {grid:>8s} dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f}")
            if n >= a.skip + a.limit:
                break
        prev_end = e


if __name__ == "__main__":
    main()
