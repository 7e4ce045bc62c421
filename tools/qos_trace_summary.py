"""Summary of a rocprofv3 --kernel-trace of tools/qos_bench: which kernels ran on which queue in each phase, and what
ran beside the slow foreground kernels."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
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
        print("  q%-3s %-48s n %6d  p50 %8.1f us  p99 %8.1f  max %8.1f" % (q, name, len(d), d[len(d) // 2], d[min(len(d) - 1, int(0.99 * len(d)))], d[-1]))
