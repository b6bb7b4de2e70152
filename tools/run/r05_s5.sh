# round 5, session 5: ties fast paths of the decode tail, NMS candidate prefetch; graph on / off; the neck configurations
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s5; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_box.py tests/test_gpu_bench_sizes.py -q -x 2>&1 | tail -15 ) > $OUT/t_box.log 2>&1; tail -5 $OUT/t_box.log
for g in 0 1; do
  timeout 400 python bench.py --steps 30 --warmup 5 --graph $g > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_g$g.json').read().strip().splitlines()[-1])
    r=d['roofline']; st=r['decode_nms_stage']
    print('graph $g', d['value'], d['ms_per_step'], d['verified'], d['stages'])
    print('   scan frac', r['frac'], 'bench stage', st['bench_input_in_line']['kernels_ms'], st['bench_input_in_line']['stage_ms'], st['bench_input_in_line']['stage_frac'])
    print('   realistic', st['realistic_heads_in_line']['kernels_ms'], st['realistic_heads_in_line']['stage_ms'], st['realistic_heads_in_line']['stage_frac'], st['realistic_heads_in_line']['scan_frac'])
    print('   heads', r['head_convs_mfma']['frac'], r['head_convs_mfma']['frac_net'], 'body', r['backbone_by_time']['ms'])
except Exception as e:
    print('graph $g FAILED', e); print(open('$OUT/bench_g$g.err').read()[-1500:])
PY
done
timeout 400 python bench.py --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 --cpu-sample 0 > $OUT/bench_fpn.json 2> $OUT/bench_fpn.err
timeout 400 python bench.py --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 --cpu-sample 0 > $OUT/bench_bifpn.json 2> $OUT/bench_bifpn.err
python - <<PY
import json
for t in ('fpn','bifpn'):
    try:
        d=json.loads(open('$OUT/bench_%s.json' % t).read().strip().splitlines()[-1])
        st=d['roofline']['decode_nms_stage']
        print(t, d['value'], d['ms_per_step'], d['verified'], 'bench stage', st['bench_input_in_line']['kernels_ms'], 'realistic', st['realistic_heads_in_line']['kernels_ms'], st['realistic_heads_in_line']['stage_frac'])
    except Exception as e:
        print(t, 'FAILED', e); print(open('$OUT/bench_%s.err' % t).read()[-800:])
PY
