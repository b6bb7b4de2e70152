# round 6, session 54: the step without a library convolution as the default -- training tests, A/B against SSDK_CONV3_NATIVE=0 at 512 / 300 px, kernel split
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s54; rm -rf $OUT; mkdir -p $OUT
( timeout 2400 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 ) > $OUT/t.log 2>&1; cat $OUT/t.log
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | cut -c60-130; }
echo default; run; echo library; SSDK_CONV3_NATIVE=0 run; echo default; run
echo "default 300"; run --size 300; echo "library 300"; SSDK_CONV3_NATIVE=0 run --size 300; echo "default 300"; run --size 300
bash tools/run/r06_evidence_train.sh > $OUT/ev.log 2>&1; grep '^{' $OUT/ev.log | cut -c60-140; grep -c -i "igemm\|miopen\|Cijk\|batched_transpose\|SubTensorOp" gpurun_out/ev_train/train_kernel_split.txt
