# round 6, session 17: phase stamps of the persistent halo kernel (tower: NHWC, head: NCHW + sigmoid), 1 tile per workgroup and one workgroup per CU
cd $GRAFT_REPO_ROOT
for n in tower_P3 tower_cls head_L1; do
  for pv in 0 1 9999; do
    for wg in 100 -1; do
      echo "== $n persist $pv wg $wg"
      SSDK_HALO_PERSIST=$pv SSDK_H3_DBG=1 SSDK_H3_DBG_WG=$wg timeout 200 python tools/gemm_probe.py $n 2>&1 | grep -E "h3p? dbg\]|TF/s" | grep -v "step:" | cut -c1-220
    done
  done
done
for n in tower_P3 tower_cls head_L1 tower_P4; do
  for pv in 0 1 9999; do
    echo "== $n persist $pv (no stamps)"
    SSDK_HALO_PERSIST=$pv timeout 200 python tools/gemm_probe.py $n 2>&1 | grep -E "TF/s" | cut -c1-200
  done
done
