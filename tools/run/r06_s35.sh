# round 6, session 35: scan unit-size sweep past one resident wave of workgroups (units that are all sample, no ring stream)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s35; mkdir -p $OUT
SWEEP_REG=16,32 SWEEP_PF=4 SWEEP_WGS=640,896,1024,1280,1536,2048,3072 timeout 600 python tools/scan_sweep.py 2>&1 | tee $OUT/sweep.txt | tail -20
