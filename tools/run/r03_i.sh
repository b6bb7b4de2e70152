cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i
mkdir -p $OUT
for cfg in "0 64" "1 64" "1 16" "4 16" "2 16"; do
set -- $cfg
echo "== SSDK_HALO_SPLITK=$1 MINP=$2" >> $OUT/gemm.log
SSDK_HALO_SPLITK=$1 SSDK_HALO_SPLITK_MINP=$2 timeout 200 python tools/gemm_probe.py head_L0 head_L1 head_L2 head_L3 >> $OUT/gemm.log 2>&1
done
grep -v amdgpu $OUT/gemm.log
