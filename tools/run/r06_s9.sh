cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $OUT
timeout 600 python tools/pw_probe.py 2>&1 | tail -27 | tee $OUT/pw_probe.txt
