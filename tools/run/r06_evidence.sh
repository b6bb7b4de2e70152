# round 6: the evidence of the three bench configurations on one box (kernel stats, PMC passes, bench lines with the oracle leg,
# per-layer tables), then the training step (bench line + kernel split).  PROF_V picks the version suffix of the files.
cd $GRAFT_REPO_ROOT
export PROF_V=${PROF_V:-1}
PROF_CFG="" bash tools/run/profile_r06.sh ev_ssd
PROF_CFG=fpn_resnet50_640 bash tools/run/profile_r06.sh ev_fpn --cfg experiments/cfgs/fpn_resnet50_640.yml --batch 32
PROF_CFG=bifpn_regnetx008_896 bash tools/run/profile_r06.sh ev_bifpn --cfg experiments/cfgs/bifpn_regnetx008_896.yml --batch 16 --dtype fp16 --graph 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/ev_train; mkdir -p $OUT
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 > $OUT/train.json 2> $OUT/train.err; tail -1 $OUT/train.json | cut -c1-300
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 --graph 1 > $OUT/train_graph.json 2>> $OUT/train.err; tail -1 $OUT/train_graph.json | cut -c1-300
timeout 400 python tools/bench_train.py --steps 30 --warmup 10 --size 300 > $OUT/train_300.json 2>> $OUT/train.err; tail -1 $OUT/train_300.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 6 --warmup 3 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/trace_tail.py $(ls $OUT/tr/*/*kernel_trace.csv | head -1) 100 400 > $OUT/train_kernel_split.txt
head -30 $OUT/train_kernel_split.txt
rm -rf $OUT/tr
