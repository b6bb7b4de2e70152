cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e
mkdir -p $OUT
for reg in 8 16; do
SSDK_SCAN_REG=$reg SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/probe_reg$reg.log 2>&1
echo "== REG=$reg"; grep -A2 "SURVEY\|nothing" $OUT/probe_reg$reg.log | grep -v "^--"
done
