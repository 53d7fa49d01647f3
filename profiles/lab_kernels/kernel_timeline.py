import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
rows.sort()
rows = [r for r in rows if "dk_" in r[2]]
# a window in the single-block region of the last step: find the last attn5 launches
idx = [i for i, r in enumerate(rows) if "attn5" in r[2]]
i0 = idx[-30]
t0 = rows[i0 - 3][0]
for s, e, k, q, g in rows[i0 - 3:i0 + 14]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{q} grid {g:>8s}  {k}")
