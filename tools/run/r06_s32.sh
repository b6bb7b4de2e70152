cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export SSDK_BN_SKIP_FINALIZE_TIMING_ONLY=1; else unset SSDK_BN_SKIP_FINALIZE_TIMING_ONLY; fi
  timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c1-160
done
