cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -40 ) > $OUT/box.log 2>&1
tail -25 $OUT/box.log
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/scan_probe.log 2>&1
tail -24 $OUT/scan_probe.log
