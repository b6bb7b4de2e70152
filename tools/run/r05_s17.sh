# round 5, session 17: halo kernel phase stamps, NHWC (tower) vs NCHW (head) epilogues
cd $GRAFT_REPO_ROOT
for n in tower_P3 head_L1; do
  for wg in 300 -1; do
    echo "== $n wg $wg"
    SSDK_H3_DBG=1 SSDK_H3_DBG_WG=$wg timeout 200 python tools/gemm_probe.py $n 2>&1 | grep -E "h3 dbg\] setup|TF/s" | cut -c1-160
  done
done
