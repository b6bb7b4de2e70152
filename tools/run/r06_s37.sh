# round 6, session 37: scan16 with a second register burst (the first 14 stream tiles of a unit requested at once while the cut is found)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s37; mkdir -p $OUT
SSDK_TAIL_STAMPS=1 timeout 600 python tools/scan_probe.py 2>&1 | grep -E "scan +[0-9]|scan16 wg0|timeline|fallback" | tee $OUT/probe.txt
for s in fpn640 bifpn896; do PROBE_SHAPE=$s timeout 600 python tools/scan_probe.py 2>&1 | grep -E "scan +[0-9]" | tee -a $OUT/probe_$s.txt; done
( timeout 1500 python -m pytest tests/test_gpu_box.py tests/test_gpu_bench_sizes.py -q -x 2>&1 | tail -4 ) | tee $OUT/t.log
