cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v7
( time timeout 500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > gpurun_out/v7/pytest.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/v7/smoke.log 2>&1
timeout 200 python bench.py --layers 1 > gpurun_out/v7/bench.json 2> gpurun_out/v7/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/v7/stats -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/v7/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/v7/pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --steps 5 > $GRAFT_REPO_ROOT/gpurun_out/v7/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/v7 -name "*_kernel_trace.csv" -size +20M -delete
ls -la gpurun_out/v7 gpurun_out/v7/stats/* | head -40
tail -3 gpurun_out/v7/pytest.log; tail -2 gpurun_out/v7/smoke.log; head -c 600 gpurun_out/v7/bench.json
