# round 6, session 50: who launches the framework's fill / copy / add kernels that are left in the training step
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s50; rm -rf $OUT; mkdir -p $OUT
timeout 600 python tools/train_glue_probe.py > $OUT/glue.txt 2> $OUT/glue.err; tail -64 $OUT/glue.txt; tail -3 $OUT/glue.err
