"""Summary of a rocprofv3 --kernel-trace of tools/qos_bench: which kernels ran on which queue in each phase, and what
ran beside the slow foreground kernels."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
if len(sys.argv) > 2:  # memory copies (rocprofv3 --memory-copy-trace): shown as "kernels" named by direction and size class
    for r in csv.DictReader(open(sys.argv[2])):
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nbytes = int(r.get("Size", r.get("Bytes", 0)) or 0)
        r["Queue_Id"] = "dma"
        r["Kernel_Name"] = "copy %s %s" % (r.get("Direction", "?"), "<64K" if nbytes < 65536 else "<4M" if nbytes < (4 << 20) else ">=4M")
        r["bytes"] = nbytes
        rows.append(r)
rows.sort(key=lambda r: r["s"])
t0 = rows[0]["s"]
# phases: gaps are rare; cut the time axis into 0.25 s windows and print per window, per (queue, kernel): n, avg, max
W = 250_000_000
wins = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    wins[(r["s"] - t0) // W][(r.get("Queue_Id", "?"), r["Kernel_Name"][:48])].append((r["e"] - r["s"]) / 1e3)
for w in sorted(wins):
    print("---- window %.2f s" % (w * W / 1e9))
    for (q, name), d in sorted(wins[w].items()):
        d.sort()
        print("  q%-3s %-48s n %6d  p50 %8.1f us  p99 %8.1f  max %8.1f  sum %9.0f us" % (q, name, len(d), d[len(d) // 2], d[min(len(d) - 1, int(0.99 * len(d)))], d[-1], sum(d)))
