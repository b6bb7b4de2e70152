# round 5, session 11: what the halo kernel's loop would gain from fewer LDS fragment reads (timing only: SSDK_H3_FAKE gives wrong results)
cd $GRAFT_REPO_ROOT
for f in 0 1 2 3; do
  echo "== SSDK_H3_FAKE=$f"
  SSDK_H3_FAKE=$f timeout 200 python tools/gemm_probe.py tower_P3 tower_P4 head_L1 2>&1 | grep -v Warn | cut -c1-110
done
