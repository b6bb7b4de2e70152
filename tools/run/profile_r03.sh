# Profiles of the default bench command on one MI355X box (run through gpurun): tools/run/profile_r03.sh <tag>
# kernel stats, the FETCH_SIZE pass, the WRITE_SIZE pass and two SQ passes, each in its own rocprofv3 run with
# --kernel-trace only. Summaries land in gpurun_out/<tag>/ ready to be copied into profiles/.
cd $GRAFT_REPO_ROOT
TAG=${1:-p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
pass() {  # name counters...
  n=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BENCH --steps 5 --warmup 2 > $OUT/$n.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python tools/pmc_to_csv.py $OUT/pmc_fetch_size.csv $OUT/fetch/*/*counter_collection.csv
python tools/pmc_to_csv.py $OUT/pmc_write_size.csv $OUT/write/*/*counter_collection.csv
python tools/pmc_to_csv.py $OUT/pmc_sq.csv $OUT/sq1/*/*counter_collection.csv $OUT/sq2/*/*counter_collection.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
head -12 $OUT/kernel_stats.csv | cut -c1-150
head -5 $OUT/pmc_sq.csv | cut -c1-300
tail -2 $OUT/sq2.log
# the bench lines themselves (CPU leg included in the first), the per-layer table, and the two neck configurations
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_layers.json 2>> $OUT/bench.err
timeout 200 python bench.py --cpu-sample 0 --graph 1 > $OUT/bench_graph.json 2>> $OUT/bench.err
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn_resnet50_640.json 2> $OUT/bench_fpn.err
timeout 300 python bench.py --cpu-sample 0 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bench_bifpn_regnetx008_896.json 2> $OUT/bench_bifpn.err
for f in bench bench_layers bench_graph bench_fpn_resnet50_640 bench_bifpn_regnetx008_896; do tail -1 $OUT/$f.json | cut -c1-330; done
tail -3 $OUT/bench_fpn.err $OUT/bench_bifpn.err
