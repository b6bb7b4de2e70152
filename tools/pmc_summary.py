"""Sums rocprofv3 counter_collection.csv per (kernel, counter): python tools/pmc_summary.py <csv> [kernel-substring]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if flt not in k:
        continue
    k = k[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k in acc:
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.0f per dispatch" % (c, v / n))
