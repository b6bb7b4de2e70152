cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BENCH > $OUT/$n.log 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS
cd $GRAFT_REPO_ROOT
python tools/pmc_to_csv.py $OUT/pmc_sq.csv $OUT/sq1/*/*counter_collection.csv $OUT/sq2/*/*counter_collection.csv
rm -rf $OUT/sq1 $OUT/sq2
grep -E "kernel,|short|halo" $OUT/pmc_sq.csv
tail -2 $OUT/sq2.log
