cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04s6; mkdir -p $OUT
for t in 0 1 2 3; do
  echo "== SSDK_TIES=$t"
  SSDK_TIES=$t timeout 300 python tools/scan_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "^SURVEY|^all equal|^trained|^uniform" | tee $OUT/probe_ties$t.txt
done
( SSDK_TIES=3 timeout 900 python -m pytest tests/test_gpu_box.py -x -q -m gpu 2>&1 | tail -4 ) > $OUT/pytest_box_ties3.log 2>&1
tail -2 $OUT/pytest_box_ties3.log
