# round 5: the oracle-checked bench lines + per-layer tables (after the PMC passes of profile_r05.sh have been copied into profiles/)
#   bash tools/run/r05_lines.sh [ssd] [fpn] [bifpn]     (default: all three)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/lines; mkdir -p $OUT
run() {  # tag args...
  t=$1; shift
  if [ -z "$*" ]; then CPU=""; else CPU="--cpu-sample 4"; fi
  timeout 600 python bench.py "$@" $CPU > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  timeout 300 python bench.py --cpu-sample 0 --layers 1 "$@" > $OUT/bench_layers_$t.json 2>> $OUT/bench_$t.err
  tail -1 $OUT/bench_$t.json | cut -c1-230
}
WHAT="${*:-ssd fpn bifpn}"
for w in $WHAT; do
  case $w in
    ssd) run ssd ;;
    fpn) run fpn --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32 ;;
    bifpn) run bifpn --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1 ;;
  esac
done
