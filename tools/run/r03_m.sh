cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu 2>&1 | tail -5 > $OUT/conv_tests.log
cat $OUT/conv_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 30 --warmup 5 > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -1 $OUT/stats.log | cut -c1-400
cut -d, -f1-4,6,7 $OUT/kernel_stats.csv | head -40 | cut -c1-200
