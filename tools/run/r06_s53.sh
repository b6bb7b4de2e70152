# round 6, session 53: small-map 3x3 layers with the batch folded into the GEMM's pixel dimension -- parity, the all-native step against the default
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_train.py -q -x -k "native_conv3x3 or head_pair_conv or whole_step" 2>&1 | grep -E "passed|failed|Error|assert|rel err" | tail -6
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130; }
for i in 1 2; do
echo default; run
echo all-native; SSDK_CONV3_NATIVE=2 run
done
