cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -30 ) > $OUT/box.log 2>&1
tail -8 $OUT/box.log
for reg in 8 16 32; do
SSDK_SCAN_REG=$reg SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/probe_reg$reg.log 2>&1
echo "== REG=$reg"; grep -A3 "SURVEY\|nothing\|all equal" $OUT/probe_reg$reg.log | grep -v "^--\|tail (kcycles)" | cut -c1-330
done
