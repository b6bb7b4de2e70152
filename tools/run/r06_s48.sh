# round 6, session 48: run-to-run scatter of the 512 px whole-step gradient case (three processes on one box)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s48; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  rm -f gpurun_out/whole_step_gradients.txt
  timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "whole_step and 512" 2>&1 | grep -E "passed|failed|plan rel|pooled" | tail -5
  cp gpurun_out/whole_step_gradients.txt $OUT/run_$i.txt
done
