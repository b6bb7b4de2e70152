# round 5, session 13: small extras + their heads as a side run next to the 8x8 head (SSDK_SSD_TAIL_SIDE=2)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s13; mkdir -p $OUT
for v in 0 2 0 2; do
  SSDK_SSD_TAIL_SIDE=$v timeout 300 python bench.py --cpu-sample 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
    print('TAIL_SIDE=$v', d['value'], d['ms_per_step'], d['verified'], d['stages'])
except Exception as e:
    print('TAIL_SIDE=$v FAILED', e); print(open('$OUT/bench_$v.err').read()[-1200:])
PY
done
