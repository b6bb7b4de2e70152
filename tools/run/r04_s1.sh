# round 4, session 1: mbflow changes (BN fold, merged stride-2 strips) + the parity / measurement holes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04s1; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "inverted_residual or stem_block or register_flow or 16x16 or mobilenetv2_plan" 2>&1 | tail -15 ) > $OUT/pytest_blocks.log 2>&1
tail -3 $OUT/pytest_blocks.log
( timeout 900 python -m pytest tests/test_gpu_bench_sizes.py tests/test_gpu_nets.py -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest_nets.log 2>&1
tail -3 $OUT/pytest_nets.log
cp gpurun_out/net_report.txt $OUT/ 2>/dev/null
timeout 400 python tools/plan_trace.py experiments/cfgs/fpn_resnet50_640.yml 32 float16 > $OUT/trace_fpn_f16.txt 2>&1
timeout 400 python tools/plan_trace.py experiments/cfgs/fpn_resnet50_640.yml 32 bfloat16 > $OUT/trace_fpn_bf16.txt 2>&1
tail -12 $OUT/trace_fpn_f16.txt
timeout 400 python bench.py --layers 1 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.err
timeout 400 python bench.py --cpu-sample 4 --layers 1 --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 > $OUT/bench_fpn.json 2> $OUT/bench_fpn.err
timeout 400 python bench.py --cpu-sample 4 --layers 1 --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 > $OUT/bench_bifpn.json 2> $OUT/bench_bifpn.err
python - <<PY
import json
for f in ["bench","bench_fpn","bench_bifpn"]:
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["verified"], r["frac"], r["decode_nms_stage"]["realistic_heads_in_line"]["stage_frac"], r.get("head_convs_mfma",{}).get("frac"), r.get("backbone_by_time",{}).get("ms"))
        print(json.dumps(d.get("forward_check"))[:1500])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 $OUT/bench_fpn.err $OUT/bench_bifpn.err
