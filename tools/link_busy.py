#!/usr/bin/env python3
"""How busy is the host link?  Reads a rocprofv3 --kernel-trace CSV and reports, for the last `frac` of the trace (the steady
state of a closed-loop run), the share of wall time during which at least one LINK kernel (gf_apply_ptrs, gf_ptrs_hash,
copy_table: the kernels that read / write caller memory) was running, and the same for all kernels.
usage: link_busy.py <kernel_trace.csv> [frac=0.3 | milliseconds]"""
import csv
import sys


def union(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    path = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
    w0 = t1 - (int(frac * 1e6) if frac > 1 else int((t1 - t0) * frac))  # a value above 1: the last so many milliseconds
    link = [(max(s, w0), e) for s, e, n in rows if e > w0 and any(x in n for x in ("gf_apply_ptrs", "gf_ptrs_hash", "copy_table"))]
    allk = [(max(s, w0), e) for s, e, n in rows if e > w0]
    names = {}
    for s, e, n in rows:
        if e > w0:
            key = n.split("(")[0][:70]
            d = names.setdefault(key, [0, 0])
            d[0] += 1
            d[1] += e - max(s, w0)
    win = t1 - w0
    print(f"window: last {win / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ms")
    print(f"link kernels running: {100.0 * union(link) / win:.1f} % of the window ({len(link)} launches)")
    print(f"any kernel running:   {100.0 * union(allk) / win:.1f} %")
    for k, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {k:70s} {c:6d} launches {d / 1e3 / max(c, 1):9.1f} us avg {100.0 * d / win:6.1f} % (sum of durations / window)")


if __name__ == "__main__":
    main()
