# round 5, session 8: the scan's near-tie rule (count the sample registers, raise the cut / settle the cut value as a tie)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s8; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_box.py -q -x 2>&1 | tail -15 ) > $OUT/t_box.log 2>&1; tail -5 $OUT/t_box.log
for sh in fpn640 ssd512; do
  ( PROBE_SHAPE=$sh SSDK_TAIL_STAMPS=1 timeout 200 python tools/scan_probe.py 2>&1 | grep -v Warn ) > $OUT/probe_$sh.log 2>&1
  echo "== $sh"; grep -E "scan  |fallback|near-tie" $OUT/probe_$sh.log | cut -c1-200 | head -40
done
timeout 400 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 0 > $OUT/bench_fpn.json 2> $OUT/bench_fpn.err
timeout 300 python bench.py --cpu-sample 0 > $OUT/bench_ssd.json 2> $OUT/bench_ssd.err
python - <<PY
import json
for t in ('fpn','ssd'):
    try:
        d=json.loads(open('$OUT/bench_%s.json' % t).read().strip().splitlines()[-1])
        st=d['roofline']['decode_nms_stage']
        print(t, d['value'], d['ms_per_step'], d['verified'], 'bench stage', st['bench_input_in_line']['kernels_ms'], 'realistic', st['realistic_heads_in_line']['kernels_ms'], st['realistic_heads_in_line']['stage_frac'])
    except Exception as e:
        print(t, 'FAILED', e); print(open('$OUT/bench_%s.err' % t).read()[-800:])
PY
