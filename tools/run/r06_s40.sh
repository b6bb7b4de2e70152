# round 6, session 40: the image-side convolution of the training step on csrc/ssdk_stemtrain.hip -- parity, whole-step gradients, A/B
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s40; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "stem_conv or whole_step" 2>&1 | grep -E "passed|failed|Error|assert|rel err|outside|dweight" | tail -8 ) > $OUT/t.log 2>&1; cat $OUT/t.log
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130; }
for i in 1 2; do
echo "stem native"; run
echo "stem library"; SSDK_STEM_NATIVE=0 run
done
echo "all native (stem + extras + every head weight gradient)"; SSDK_CONV3_NATIVE=2 SSDK_HEAD_WGRAD_MIN=0 run
