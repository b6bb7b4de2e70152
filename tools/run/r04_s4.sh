# round 4, session 4: LDS layouts of the head kernels for the real ds_read_b128 lane groups -- parity, A/B, conflict counters
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s4; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest_conv.log 2>&1
tail -3 $OUT/pytest_conv.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
    h=d["roofline"]["head_convs_mfma"]
    print("$tag", d["value"], d["ms_per_step"], "heads", h["ms"], h["frac"], " ".join("%s:%.1f" % (r["kernel"][:12], r["us"]) for r in d["layers"] if r["kind"]=="head" or "smallmap" in r["kernel"] or "xpair" in r["kernel"]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run new
run old SSDK_S3_PAD=16 SSDK_H3_SWZ=0
run new2
BENCH="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_to_csv.py $OUT/pmc_sq.csv $OUT/sq/*/*counter_collection.csv
rm -rf $OUT/sq
grep -E "conv|xpair|kernel,disp" $OUT/pmc_sq.csv | cut -c1-260 | head -14
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn.json 2> $OUT/bench_fpn.err
SSDK_S3_PAD=16 SSDK_H3_SWZ=0 timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn_old.json 2> $OUT/bench_fpn_old.err
python - <<PY
import json
for f in ["bench_fpn","bench_fpn_old"]:
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], r.get("head_convs_mfma",{}).get("frac"), r.get("head_convs_mfma",{}).get("ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
