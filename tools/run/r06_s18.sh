# round 6, session 18: GEMM yardstick (library, random vs zero operands) + the halo kernel on zero operands
cd $GRAFT_REPO_ROOT
timeout 600 python tools/gemm_ceiling_probe.py 2>&1 | grep -v Warning
echo "== halo kernel, zero operands (SSDK_PROBE_ZERO=1)"
for n in tower_P3 tower_cls head_L1; do
  SSDK_PROBE_ZERO=1 timeout 200 python tools/gemm_probe.py $n 2>&1 | grep -E "TF/s" | cut -c1-200
  timeout 200 python tools/gemm_probe.py $n 2>&1 | grep -E "TF/s" | cut -c1-200
done
