#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes written by scripts/pmc_lab.sh / gpu_round.sh pmc:
per kernel, counter sums averaged per dispatch, with kernel duration from the kernel trace.
usage: python scripts/pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if filt in k:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if filt in k:
                durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in vals:
        d = durs.get(k, [0.0])
        print(f"## {k[:100]}\n  dispatches/pass ~{len(d) // max(1, len(glob.glob(os.path.join(root, 'p*/'))))}  duration us: "
              f"avg {sum(d) / len(d):.1f} min {min(d):.1f}")
        for c, v in sorted(vals[k].items()):
            print(f"  {c:28s} per dispatch {sum(v) / len(v):.6g}   (n={len(v)})")


if __name__ == "__main__":
    main()
