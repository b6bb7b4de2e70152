# round 5, session 10: the three bench configurations (oracle-checked) on the scan with the rank rule / shared key buffer / near-tie rule
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s10; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_bench_sizes.py -q -x 2>&1 | tail -5 ) > $OUT/t_box.log 2>&1; tail -3 $OUT/t_box.log
timeout 400 python bench.py > $OUT/bench_ssd.json 2> $OUT/bench_ssd.err
timeout 600 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 4 > $OUT/bench_fpn.json 2> $OUT/bench_fpn.err
timeout 600 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 --cpu-sample 4 > $OUT/bench_bifpn.json 2> $OUT/bench_bifpn.err
python - <<PY
import json
for t in ('ssd','fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/bench_%s.json' % t).read().strip().splitlines()[-1])
        st=d['roofline']['decode_nms_stage']
        print(t, d['value'], d['ms_per_step'], d['verified'], 'bench stage', st['bench_input_in_line']['kernels_ms'], st['bench_input_in_line']['stage_frac'], 'realistic', st['realistic_heads_in_line']['kernels_ms'], st['realistic_heads_in_line']['stage_frac'], st['realistic_heads_in_line']['scan_frac'])
    except Exception as e:
        print(t, 'FAILED', e); print(open('$OUT/bench_%s.err' % t).read()[-800:])
PY
