# round 5, session 7: plan ordering experiment (extras + small heads as a side-stream run), decode tail stream, scan probe on near ties
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s7; mkdir -p $OUT
for v in "X=0" "SSDK_SSD_TAIL_SIDE=1" "X=0 TAIL=1" "SSDK_SSD_TAIL_SIDE=1 TAIL=1"; do
  tag=$(echo $v | tr '= ' '__')
  extra=""; case "$v" in *TAIL=1*) extra="--tail-stream 1";; esac
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --cpu-sample 0 $extra > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1])
    print('$v', d['value'], d['ms_per_step'], d['verified'], d['stages'])
except Exception as e:
    print('$v', 'FAILED', e); print(open('$OUT/bench_$tag.err').read()[-600:])
PY
done
( PROBE_SHAPE=fpn640 SSDK_TAIL_STAMPS=1 timeout 200 python tools/scan_probe.py 2>&1 | grep -v Warn ) > $OUT/probe_fpn.log 2>&1; grep -A12 "near ties" $OUT/probe_fpn.log | head -16
