# round 6, session 43: stem kernels at 300 px (forward ours, weight gradient of that width on the library) -- parity, A/B
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "stem_conv" 2>&1 | grep -E "passed|failed|Error|assert|rel err|outside|dweight" | tail -8 )
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | cut -c60-130; }
for i in 1 2 3; do
echo "300 px stem native"; run --size 300
echo "300 px stem library"; SSDK_STEM_NATIVE=0 run --size 300
done
