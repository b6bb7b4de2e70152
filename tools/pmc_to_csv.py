"""Aggregates one or more rocprofv3 `*_counter_collection.csv` files into one row per kernel:
    python tools/pmc_to_csv.py out.csv pass1.csv [pass2.csv ...]
Every counter becomes a column holding its per-dispatch mean (summed over XCDs/SEs, as rocprofv3 reports it).
FETCH_SIZE additionally gets the `bytes_per_dispatch_x2_gfx950_correction` column bench.py reads: the counter is in KB
and, on gfx950, counts 64 B per 128-B request of a wide coalesced read (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is
NOT corrected: the guide calibrates the x2 for wide reads only, and the raw figure of a known write (the stem block's
134 217 728-byte output reads 131 072 KB) already equals the algorithmic bytes -- `write_bytes_per_dispatch` = KB x 1024."""
import collections
import csv
import sys

out, srcs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for s in srcs:
    for r in csv.DictReader(open(s)):
        k = r["Kernel_Name"].split("(")[0].strip()
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
cols = ["kernel", "dispatches"] + names
if "FETCH_SIZE" in names:
    cols.append("bytes_per_dispatch_x2_gfx950_correction")
if "WRITE_SIZE" in names:
    cols.append("write_bytes_per_dispatch")
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(cols)
    order = sorted(acc, key=lambda k: -max(acc[k].values()))
    for k in order:
        row = [k, max(len(v) for v in cnt[k].values())]
        for c in names:
            n = len(cnt[k][c])
            row.append(round(acc[k][c] / n) if n else "")
        if "FETCH_SIZE" in names:
            n = len(cnt[k]["FETCH_SIZE"])
            row.append(round(acc[k]["FETCH_SIZE"] / n * 1024 * 2) if n else "")
        if "WRITE_SIZE" in names:
            n = len(cnt[k]["WRITE_SIZE"])
            row.append(round(acc[k]["WRITE_SIZE"] / n * 1024) if n else "")
        w.writerow(row)
