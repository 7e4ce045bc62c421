#!/usr/bin/env python3
"""Summarise rocprofv3 `--pmc ... --output-format csv` counter_collection.csv files:
per (kernel, counter) the per-dispatch average (counter values are summed over
the dimension rows of one dispatch first).

usage: pmc_summary.py <dir-or-csv> [<dir-or-csv> ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name: str, n: int = 72) -> str:
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main() -> None:
    for arg in sys.argv[1:]:
        files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "*counter_collection.csv"))
        for f in files:
            per = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch -> value
            dur = defaultdict(list)
            for r in csv.DictReader(open(f)):
                key = (r["Kernel_Name"], r["Counter_Name"])
                per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
                dur[(r["Kernel_Name"], r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print(f"# {f}")
            print(f"{'counter':>24} {'dispatches':>10} {'avg/dispatch':>18} {'avg_dur_us':>11}  kernel")
            for (kn, cn), d in sorted(per.items()):
                if "at::native" in kn or "rocclr" in kn:
                    continue
                vals = list(d.values())
                ds = [dur[(kn, i)] for i in d]
                print(f"{cn:>24} {len(vals):>10} {sum(vals)/len(vals):>18.1f} {sum(ds)/len(ds)/1e3:>11.1f}  {short(kn)}")


if __name__ == "__main__":
    main()
