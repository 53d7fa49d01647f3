"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace (csv): python scripts/launch_gaps.py <dir>.
Gaps between dk_ kernels of the denoising loop (a gap > 100 us is a host stall between images / phases and is listed apart)."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = [r for r in rows if "dk_" in r[2]]
busy = sum(e - s for s, e, k in rows)
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g[0] < 100_000]
neg = [g for g in gaps if g[0] < 0]
print(f"{len(rows)} dk kernels, busy {busy / 1e6:.2f} ms; {len(small)} gaps < 100 us: total {sum(g[0] for g in small) / 1e6:.3f} ms, mean {sum(g[0] for g in small) / max(len(small), 1) / 1e3:.2f} us, "
      f"median {sorted(g[0] for g in small)[len(small) // 2] / 1e3:.2f} us; overlapping pairs {len(neg)}; larger gaps {len(gaps) - len(small) - len(neg)}")
by = {}
for g, a, b in small:
    key = (a.split("(")[0][-28:], b.split("(")[0][-28:])
    by.setdefault(key, []).append(g)
for key, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {key[0]:>28s} -> {key[1]:<28s} n {len(v):5d} mean {sum(v) / len(v) / 1e3:6.2f} us total {sum(v) / 1e6:7.3f} ms")
