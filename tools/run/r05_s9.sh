# round 5, session 9: scan timeline on the FPN configuration's own bench input
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $OUT
( PROBE_SHAPE=fpn640 PROBE_CFG=experiments/cfgs/fpn_resnet50_640.yml SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v Warn ) > $OUT/probe_fpn.log 2>&1
head -12 $OUT/probe_fpn.log | cut -c1-400
( timeout 900 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -5 ) > $OUT/t_box.log 2>&1; tail -3 $OUT/t_box.log
( PROBE_SHAPE=ssd512 SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v Warn | grep -E "scan  " | cut -c1-200 ) 
( PROBE_SHAPE=fpn640 SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py 2>&1 | grep -v Warn | grep -E "scan  |near-tie" | cut -c1-260 ) 
