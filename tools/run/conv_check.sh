# conv parity tests + per-layer bench table (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "not soak" 2>&1 | tail -3
timeout 200 python bench.py --layers 1 --cpu-sample 0 > gpurun_out/mb_bench.json 2>gpurun_out/mb_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/mb_bench.json'))
print(d['value'], d['ms_per_step'], d['stages'], d.get('verified'), d['roofline'].get('head_convs_mfma'))
for r in d['layers']:
    if r['kind'] != 'mbconv': print("%-40s %-34s %7.1f %7.1f TF" % (r['layer'], r['kernel'], r['us'], r['TFLOPs']))
PY
