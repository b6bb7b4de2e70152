# mbconv parity tests + per-layer bench table (run through gpurun)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_conv.py -q -x -k "inverted or stem or 16x16 or mobilenetv2" 2>&1 | tail -3
timeout 200 python bench.py --layers 1 --cpu-sample 0 > gpurun_out/mb_bench.json 2>gpurun_out/mb_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/mb_bench.json'))
print(d['value'], d['ms_per_step'], d['stages'], d.get('verified'))
for r in d['layers']:
    print("%-40s %-34s %7.1f" % (r['layer'], r['kernel'], r['us']))
PY
