cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s8; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_conv.py tests/test_gpu_nets.py -q -m gpu -k "(forward_at_bench_size and ssd_mobilenetv2_512 and bfloat16) or (large_tile and 64-256-1-1-64-64-8) or (plan_matches_reference_module and fpn_r50 and bfloat16)" 2>&1 | grep -v "^  " | tail -150 ) > $OUT/fail.log 2>&1; tail -150 $OUT/fail.log
