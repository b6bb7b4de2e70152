# round 5, session 15: what fewer image loads in the stem block would buy (timing only: variant libraries that load N of the 8 values per lane)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s15; mkdir -p $OUT
cp ssds.pytorch_amd/csrc/libssdk.so /tmp/libssdk_keep.so
for v in 8 5 3 8 3; do
  cp tmp_libs/libssdk_n$v.so ssds.pytorch_amd/csrc/libssdk.so
  timeout 300 python bench.py --cpu-sample 0 --layers 1 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
    print('N=$v', d['value'], d['ms_per_step'], [ (r['kernel'], r['us']) for r in d['layers'][:3]])
except Exception as e:
    print('N=$v FAILED', e); print(open('$OUT/bench_$v.err').read()[-800:])
PY
done
cp /tmp/libssdk_keep.so ssds.pytorch_amd/csrc/libssdk.so
