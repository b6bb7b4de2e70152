cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03g
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -30 ) > $OUT/box.log 2>&1
tail -5 $OUT/box.log
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/probe.log 2>&1
grep -A3 "SURVEY\|nothing\|all equal" $OUT/probe.log | grep -v "^--" | cut -c1-330
