#!/usr/bin/env python
"""HBM traffic per launch of every kernel from the FETCH_SIZE / WRITE_SIZE passes written by
`scripts/gpu_round.sh <tag> pmc` (rocprofv3 --pmc, CSV).  Corrections per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane
global_load and buffer_load..lds alike), so it is doubled; WRITE_SIZE is taken as is (uncalibrated).
Writes <dir>/pmc_gemm_traffic.json (to be copied to profiles/ for bench.py's roofline.traffic) and prints a
markdown table.
usage: python scripts/pmc_traffic.py gpurun_out/<tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(root, counter):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals


def main():
    root = sys.argv[1]
    fetch, write = load(root, "FETCH_SIZE"), load(root, "WRITE_SIZE")
    rows = []
    for k in fetch:
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        rows.append((k, len(fetch[k]), f, w, (2 * f + w) * 1024))
    rows.sort(key=lambda r: -r[4] * r[1])
    print("| kernel | launches | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM bytes/launch (2*fetch + write) |\n|---|---|---|---|---|")
    for k, n, f, w, b in rows[:14]:
        print(f"| `{k[:70]}` | {n} | {f:.0f} | {w:.0f} | {b:.4g} |")
    # the dominant GEMM kernel of the run = the GEMM kernel with the most bytes over all its launches (rows are sorted that way)
    dom = [r for r in rows if "gemm256" in r[0]] or [r for r in rows if "gemm" in r[0]]
    if dom:
        k, n, f, w, b = dom[0]
        out = {"kernel": k, "launches": n, "fetch_kib_per_launch_raw": f, "write_kib_per_launch": w, "hbm_bytes_per_launch": b,
               "correction": "FETCH_SIZE doubled (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
               "source": root}
        dst = os.path.join(root, "pmc_gemm_traffic.json")  # copy to profiles/pmc_gemm_traffic.json (only gpurun_out/ travels back)
        json.dump(out, open(dst, "w"), indent=1)
        print("\nwrote", dst, json.dumps(out)[:300])


if __name__ == "__main__":
    main()
