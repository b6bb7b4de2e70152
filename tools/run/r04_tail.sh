cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04tail; mkdir -p $OUT; rm -f gpurun_out/net_report.txt
for v in "X=1" "SSDK_MB_FLOW=0 SSDK_MB_SPLIT=0" "SSDK_MB_SPLIT=0"; do
  echo "== $v" >> gpurun_out/net_report.txt
  env $v timeout 300 python -m pytest tests/test_gpu_nets.py -q -m gpu -k "plan_matches and ssd_mnv2" 2>&1 | tail -3
done
grep -A14 "^==" gpurun_out/net_report.txt | grep "==\|bfloat16\|float16\|conf0\|conf1\|loc0"
