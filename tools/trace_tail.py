"""Per-kernel totals over the last <ms> milliseconds of a rocprofv3 kernel trace (one steady-state step)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45  # kernels listed
end = max(int(r["End_Timestamp"]) for r in rows)
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= end - ms * 1e6:
        k = r["Kernel_Name"][:90]
        acc[k][0] += 1
        acc[k][1] += (e - s) / 1e3
        tot += (e - s) / 1e3
for k, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f us %5.1f%% x%-4d %s" % (us, 100 * us / tot, n, k))
print("kernel time in window: %.1f ms" % (tot / 1e3))
