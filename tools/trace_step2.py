"""One step of a rocprofv3 kernel trace with absolute start/end (us from the step's first kernel) per stream."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nms = [i for i, r in enumerate(rows) if "nms_kernel" in r["Kernel_Name"]]
a, b = nms[len(nms) // 2 - 1] + 1, nms[len(nms) // 2] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ssdk::", "")[:44]
    print("%-46s q%-3s %9.1f -> %9.1f  (%7.1f us)" % (name, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
