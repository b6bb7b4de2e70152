# round 6, session 46: the whole-step gradient test at 512 px alone, with its assertion text
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_train.py -q -x -k "whole_step and 512" 2>&1 | grep -v "^$" | tail -40
