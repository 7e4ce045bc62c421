#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (ROCm 7.x default output) into the
compact per-kernel text summary committed under profiles/.

usage: rocprof_summary.py <results.db> [--pmc]   (prints to stdout)
"""
import sqlite3
import sys


def short(name: str, n: int = 96) -> str:
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print(f"# source: {sys.argv[1]}")
    print("# kernel-trace summary (durations in microseconds)")
    print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_us':>12} {'%':>6}  kernel [grid x wg, vgpr, sgpr, lds]")
    rows = cur.execute(
        "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[5] for r in rows) or 1
    for name, calls, avg, mn, mx, tot, gx, wx, vg, sg, lds in rows:
        print(f"{calls:>6} {avg/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {tot/1e3:>12.1f} {100*tot/total:>6.2f}  "
              f"{short(name)} [{gx}x{wx}, v{vg}, s{sg}, lds{lds}]")
    if "--pmc" in sys.argv:
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:
            print("\n# PMC counters (sum over dispatches / per dispatch average)")
            q = ("select kernel_name, counter_name, count(*), sum(value), avg(value) from "
                 "(select k.name as kernel_name, c.counter_name as counter_name, c.dispatch_id as d, sum(c.value) as value "
                 " from counters_collection c join kernels k on k.dispatch_id = c.dispatch_id "
                 " group by k.name, c.counter_name, c.dispatch_id) group by kernel_name, counter_name")
            try:
                for kn, cn, n, s, a in cur.execute(q):
                    print(f"{cn:>28} dispatches={n:<5} per_dispatch={a:>16.1f}  {short(kn, 70)}")
            except sqlite3.Error as e:
                print("could not read counters:", e)


if __name__ == "__main__":
    main()
