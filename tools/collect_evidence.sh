# copies the summaries of tools/run/r06_evidence.sh from gpurun_out/ into profiles/ under the round's names:  bash tools/collect_evidence.sh <version>
V=${1:-1}
cd "$(dirname "$0")/.."
for pair in "ev_ssd:" "ev_fpn:_fpn_resnet50_640" "ev_bifpn:_bifpn_regnetx008_896"; do
  d=gpurun_out/${pair%%:*}; s=${pair##*:}
  [ -d $d ] || continue
  cp $d/kernel_stats.csv profiles/r06_bench${s}_kernel_stats_v$V.csv
  cp $d/pmc_fetch_size.csv profiles/r06_pmc_fetch_size${s}_v$V.csv
  cp $d/pmc_write_size.csv profiles/r06_pmc_write_size${s}_v$V.csv
  cp $d/pmc_sq.csv profiles/r06_pmc_sq${s}_v$V.csv
  tail -1 $d/bench.json > profiles/r06_bench${s}_v$V.json
  tail -1 $d/bench_layers.json > profiles/r06_bench_layers${s}_v$V.json
done
if [ -d gpurun_out/ev_train ]; then
  ( tail -1 gpurun_out/ev_train/train.json; tail -1 gpurun_out/ev_train/train_graph.json; tail -1 gpurun_out/ev_train/train_300.json ) > profiles/r06_train_step_final_v$V.json
  cp gpurun_out/ev_train/train_kernel_split.txt profiles/r06_train_kernel_split_final_v$V.txt
fi
ls -la profiles | grep "r06_" | wc -l
