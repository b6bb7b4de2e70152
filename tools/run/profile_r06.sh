# Round-6 profiles of ONE bench configuration on one MI355X box (run through gpurun):
#   bash tools/run/profile_r06.sh <tag> [bench.py arguments of the configuration ...]
# kernel stats, the FETCH_SIZE pass, the WRITE_SIZE pass and two SQ passes, each in its own rocprofv3 run with --kernel-trace
# only; then the bench lines themselves (with the CPU leg / oracle check: --cpu-sample 4 for the neck configurations) and the
# per-layer table.  Summaries land in gpurun_out/<tag>/ ready to be copied into profiles/.
cd $GRAFT_REPO_ROOT
TAG=${1:-p}; shift
ARGS="$@"
ARGS_ABS=$(echo "$@" | sed "s#experiments/#$GRAFT_REPO_ROOT/experiments/#g")  # (the profiler passes run from /tmp)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 $ARGS_ABS"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
pass() {  # name counters...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BENCH --steps 5 --warmup 2 > $OUT/$n.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU
cd $GRAFT_REPO_ROOT
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python tools/pmc_to_csv.py $OUT/pmc_fetch_size.csv $OUT/fetch/*/*counter_collection.csv
python tools/pmc_to_csv.py $OUT/pmc_write_size.csv $OUT/write/*/*counter_collection.csv
python tools/pmc_to_csv.py $OUT/pmc_sq.csv $OUT/sq1/*/*counter_collection.csv $OUT/sq2/*/*counter_collection.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
head -8 $OUT/kernel_stats.csv | cut -c1-150
# the bench lines below quote the rocprof figures of THIS command: put the summaries where bench.py looks for them
# (profiles/r06_bench[_<cfg>]_kernel_stats_v<N>.csv, profiles/r06_pmc_fetch_size[_<cfg>]_v<N>.csv; PROF_CFG / PROF_V from the caller)
V=${PROF_V:-1}
if [ -n "$PROF_CFG" ]; then SFX="_$PROF_CFG"; else SFX=""; fi
cp $OUT/kernel_stats.csv profiles/r06_bench${SFX}_kernel_stats_v$V.csv
cp $OUT/pmc_fetch_size.csv profiles/r06_pmc_fetch_size${SFX}_v$V.csv
cp profiles/r06_bench${SFX}_kernel_stats_v$V.csv profiles/r06_pmc_fetch_size${SFX}_v$V.csv $OUT/
[ -n "$PROFILE_ONLY" ] && exit 0
if [ -z "$ARGS" ]; then CPU=""; else CPU="--cpu-sample 4"; fi
timeout 600 python bench.py $ARGS $CPU > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --cpu-sample 0 --layers 1 $ARGS > $OUT/bench_layers.json 2>> $OUT/bench.err
for f in bench bench_layers; do tail -1 $OUT/$f.json | cut -c1-400; done
tail -2 $OUT/bench.err
