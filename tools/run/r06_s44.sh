# round 6, session 44: run-to-run spread of the training step on one box (512 px eager, 300 px)
cd $GRAFT_REPO_ROOT
run() { timeout 400 python tools/bench_train.py --steps 30 --warmup 10 "$@" 2>/dev/null | tail -1 | cut -c60-130; }
for i in 1 2 3; do run; done
for i in 1 2 3 4; do run --size 300; done
