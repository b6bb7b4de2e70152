# round 6, session 39: what the remaining library convolutions cost -- extras' 3x3 layers on the native path, small-level head weight gradients native
cd $GRAFT_REPO_ROOT
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 2>/dev/null | tail -1 | cut -c60-130; }
for i in 1 2; do
echo "default"; run
echo "CONV3_NATIVE=2 (extras)"; SSDK_CONV3_NATIVE=2 run
echo "both"; SSDK_CONV3_NATIVE=2 SSDK_HEAD_WGRAD_MIN=0 run
done
