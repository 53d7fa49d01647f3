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
    # the block Linears' GEMM kernels of the run (bench.py's roofline class 0: gemm256v4 + gemm256v3 instantiations), launch-weighted
    dom = [r for r in rows if "gemm256v" in r[0]] or [r for r in rows if "gemm" in r[0]]
    if dom:
        n_all = sum(r[1] for r in dom)
        f = sum(r[2] * r[1] for r in dom) / n_all
        w = sum(r[3] * r[1] for r in dom) / n_all
        b = sum(r[4] * r[1] for r in dom) / n_all
        out = {"kernel": " + ".join(f"{r[0].split('(')[0].replace('void ', '')} ({r[1]} launches)" for r in dom) + ", launch-weighted",
               "launches": n_all, "fetch_kib_per_launch_raw": f, "write_kib_per_launch": w, "hbm_bytes_per_launch": b,
               "per_instantiation": {r[0].split("(")[0].replace("void ", ""): {"launches": r[1], "hbm_bytes_per_launch": r[4]} for r in dom},
               "correction": "FETCH_SIZE doubled (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
               "source": root}
        dst = os.path.join(root, "pmc_gemm_traffic.json")  # copy to profiles/pmc_gemm_traffic.json (only gpurun_out/ travels back)
        json.dump(out, open(dst, "w"), indent=1)
        print("\nwrote", dst, json.dumps(out)[:400])


if __name__ == "__main__":
    main()
