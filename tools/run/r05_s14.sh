# round 5, session 14: the scan with 32 sample tiles (SSDK_SCAN_REG=32) on the FPN configuration's own bench input and on realistic heads
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s14; mkdir -p $OUT
for reg in 16 32; do
  echo "== SSDK_SCAN_REG=$reg"
  ( SSDK_SCAN_REG=$reg PROBE_SHAPE=fpn640 PROBE_CFG=experiments/cfgs/fpn_resnet50_640.yml SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v Warn ) > $OUT/probe_fpn_$reg.log 2>&1
  grep -E "scan  |near-tie|fallback" $OUT/probe_fpn_$reg.log | cut -c1-250 | head -12
  ( SSDK_SCAN_REG=$reg PROBE_SHAPE=ssd512 timeout 300 python tools/scan_probe.py 2>&1 | grep -v Warn | grep -E "scan  " | cut -c1-160 | head -3 )
done
