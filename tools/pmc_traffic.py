#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic per launch.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, so wide coalesced reads are doubled;
WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.  The two counters do not fit one pass.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv \
        gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_pmc_traffic.json
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


def collect(path, counter):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    fa, wa = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fa) | set(wa)):
        f, w = fa.get(k, [0, 0, 0]), wa.get(k, [0, 0, 0])
        n = max(f[1], w[1])
        rd = 2.0 * f[0] * 1024 / max(1, f[1])       # gfx950: FETCH_SIZE reports half of wide streaming reads
        wr = w[0] * 1024 / max(1, w[1])
        rows[k] = dict(launches=n, read_bytes_per_launch=round(rd), write_bytes_per_launch=round(wr),
                       hbm_bytes_per_launch=round(rd + wr), avg_us_under_pmc=round(f[2] / max(1, f[1]), 2))
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows.values())
    doc = dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh: bench.py --steps 1 "
                    "--warmup 0 --global-batch <one micro-batch>; the headline config: 64 clips); FETCH_SIZE x2 (gfx950 correction), KiB -> bytes",
               total_hbm_bytes=tot, kernels=rows)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    top = sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]
    for k, v in top:
        print(f"{k[:70]:70s} n={v['launches']:4d} rd={v['read_bytes_per_launch']/1e6:9.2f}MB wr={v['write_bytes_per_launch']/1e6:9.2f}MB")
    print("total GB:", tot / 1e9)


if __name__ == "__main__":
    main()
