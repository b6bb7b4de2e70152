"""Prints one step (nms -> nms) of a rocprofv3 kernel trace: kernel, duration, gap to the previous kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nms = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
a, b = nms[len(nms) // 2 - 1] + 1, nms[len(nms) // 2] + 1
prev = int(rows[a - 1]["End_Timestamp"])
tot = gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ssdk::", "")[:48]
    print("%-50s %8.1f us  gap %6.1f" % (name, (e - s) / 1e3, (s - prev) / 1e3))
    tot += e - s
    gap += s - prev
    prev = e
print("kernels %.1f us, gaps %.1f us" % (tot / 1e3, gap / 1e3))
