cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03l
mkdir -p $OUT
for nw in 4 2 1; do
echo "== NW=$nw" >> $OUT/gemm.log
SSDK_CONV_SMALLMAP_NW=$nw timeout 200 python tools/gemm_probe.py head_L2 head_L3 head_L4 head_L5 >> $OUT/gemm.log 2>&1
done
grep -v amdgpu $OUT/gemm.log
