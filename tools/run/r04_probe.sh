cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04probe; mkdir -p $OUT
SSDK_TAIL_STAMPS=1 timeout 300 python tools/scan_probe.py > $OUT/scan_probe.txt 2>&1
cat $OUT/scan_probe.txt | grep -v amdgpu.ids
