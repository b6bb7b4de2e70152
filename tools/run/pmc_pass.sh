# FETCH_SIZE pass + default bench line (run through gpurun)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/v7b
mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob
src = glob.glob('gpurun_out/v7b/pmc/*/*_counter_collection.csv')[0]
acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0]
    acc[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
with open('profiles/r01_pmc_fetch_size_v7.csv', 'w') as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,bytes_per_dispatch_x2_gfx950_correction\n")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        n = len(cnt[k]); per = v / n
        f.write('"%s",%d,%d,%d\n' % (k, n, round(per), round(per * 1024 * 2)))
PY
cp profiles/r01_pmc_fetch_size_v7.csv $OUT/
grep scan profiles/r01_pmc_fetch_size_v7.csv
timeout 200 python bench.py --layers 1 > $OUT/bench.json 2>/dev/null
head -c 250 $OUT/bench.json
