cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03h
mkdir -p $OUT
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/probe.log 2>&1
grep -A3 "SURVEY\|all equal" $OUT/probe.log | grep -v "^--" | cut -c1-330
