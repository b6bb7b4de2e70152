cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s9; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bench_sizes.py tests/test_gpu_nets.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/t.log 2>&1; tail -25 $OUT/t.log
for c in "" "--cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32" "--cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16"; do
  timeout 300 python bench.py --cpu-sample 0 $c 2>/dev/null | tail -1 | cut -c1-200
done
