#!/usr/bin/env python
"""Per-kernel statistics (calls, total / avg / min / max duration, share) from a rocprofv3
`--kernel-trace --stats` run.  rocprofv3 7.x writes a rocpd SQLite database (`*_results.db`); this
prints the same table its CSV `kernel_stats` would hold, as markdown, for `profiles/`.

    python scripts/rocpd_summary.py gpurun_out/<tag>/prof/flux_results.db [--pmc] > profiles/<name>.md
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"source: `{path}` (rocprofv3 --kernel-trace --stats, rocpd database)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | max grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx, vg, ag, lds, grid, wg in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} "
              f"| {vg} | {ag} | {lds} | {grid} | {wg} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if "--by-grid" in sys.argv:
        rows = cur.execute(
            "select name, grid_x, count(*), sum(duration), avg(duration) from kernels where name like '%gemm%' or name like '%attn%' or name like '%conv%' or name like '%gn_%' "
            "group by name, grid_x order by sum(duration) desc").fetchall()
        print("\n| kernel | grid_x (threads) | calls | total ms | avg us |\n|---|---|---|---|---|")
        for name, grid, n, tot, avg in rows[:40]:
            print(f"| `{name[:60]}` | {grid} | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} |")
    if "--pmc" in sys.argv:
        try:
            pr = cur.execute("select k.name, p.name, count(*), sum(e.value) from pmc_events e "
                             "join kernels k on k.dispatch_id = e.dispatch_id join pmc_info p on p.id = e.pmc_id "
                             "group by k.name, p.name order by sum(e.value) desc").fetchall()
            print("\n| kernel | counter | dispatches | sum | per dispatch |\n|---|---|---|---|---|")
            for k, c, n, s in pr:
                print(f"| `{k[:80]}` | {c} | {n} | {s:.6g} | {s / max(n, 1):.6g} |")
        except sqlite3.Error as e:  # schema differs between rocprofv3 builds
            print(f"\n(pmc tables not readable: {e})")


if __name__ == "__main__":
    main()
