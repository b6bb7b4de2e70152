# round 4, session 3: split kernel -- dispatch fix, taps in registers (XB=3), segment rule; SQ counters of the block kernels
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s3; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "split_register or inverted_residual or register_flow" 2>&1 | tail -15 ) > $OUT/pytest_blocks.log 2>&1
tail -4 $OUT/pytest_blocks.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_$tag.json") if l.startswith("{")][-1])
    print("$tag", d["value"], d["ms_per_step"], " ".join("%s:%.1f" % (r["kernel"][:10], r["us"]) for r in d["layers"][:8]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run xb1 SSDK_MB_SPLIT_XB=1
run treg SSDK_MB_SPLIT_XB=3
run treg_min350 SSDK_MB_SPLIT_XB=3 SSDK_MB_SPLIT_MIN=350
run xb1_min350 SSDK_MB_SPLIT_XB=1 SSDK_MB_SPLIT_MIN=350
BENCH="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BENCH > $OUT/$n.log 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES
cd $GRAFT_REPO_ROOT
python tools/pmc_to_csv.py $OUT/pmc_sq.csv $OUT/sq1/*/*counter_collection.csv $OUT/sq2/*/*counter_collection.csv
rm -rf $OUT/sq1 $OUT/sq2
grep -E "mbsplit|mbflow|kernel" $OUT/pmc_sq.csv | cut -c1-400 | head -20
tail -2 $OUT/sq2.log
