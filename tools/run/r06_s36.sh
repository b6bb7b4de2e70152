# round 6, session 36: phase stamps + workgroup timeline of scan16 on the SSD shape (where the per-unit fixed cost sits)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s36; mkdir -p $OUT
SSDK_TAIL_STAMPS=1 timeout 600 python tools/scan_probe.py 2>&1 | tee $OUT/probe.txt | tail -60
