import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["stages"])
tot={}
for l in d["layers"]:
    tot[l["kind"]]=tot.get(l["kind"],0)+l["us"]
    if l["kind"]!="mbconv": print("%-34s %-14s %7.1f us %7.1f TF" % (l["layer"], l["kernel"], l["us"], l["TFLOPs"]))
print(tot)
