# round 6, session 22: SSD head pairs of the training step on the inference kernels (SSDK_HEAD_PAIR) -- parity, whole-step gradients, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s22; mkdir -p $OUT
( timeout 1500 python -m pytest tests/test_gpu_train.py -q -x -k "head_pair or whole_step or training_module or graphed" 2>&1 | tail -12 ) > $OUT/t_train.log 2>&1; tail -12 $OUT/t_train.log
for v in 1 0 1 0; do
  SSDK_HEAD_PAIR=$v timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train_hp$v.json 2> $OUT/train_hp$v.err
  tail -1 $OUT/train_hp$v.json | cut -c1-300
done
