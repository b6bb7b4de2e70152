cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 600 python tools/scan_sweep.py > $OUT/sweep.log 2>&1
tail -30 $OUT/sweep.log
