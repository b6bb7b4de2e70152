cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03j
mkdir -p $OUT
SSDK_H3_DBG=1 timeout 200 python tools/gemm_probe.py head_L0 head_L1 tower_P3 > $OUT/gemm.log 2>&1
grep -v amdgpu $OUT/gemm.log | grep -v "step: issue"
