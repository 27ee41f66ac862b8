# rocprofv3 evidence for one of the other BASELINE configurations (tools/bench_config.py): kernel stats of the synchronous call
# and of the lanes, VALU-issue / HBM-byte counter passes of the synchronous call (separate --pmc passes, as the guide prescribes),
# summarised into profiles/<tag>_<config>_*.  usage: bash tools/prof_config.sh r03 c5 [extra bench_config.py flags]
set -e
TAG=$1; CFG=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG}_$CFG; mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
cd $R
P="python3 tools/bench_config.py --config $CFG $*"
python3 tools/bench_config.py --config $CFG --bf $* > $OUT/rate.json 2> $OUT/rate.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P --mode sync --steps 10 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes -- $P --mode lanes --steps 30 > $OUT/stats_lanes.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq -- $P --mode sync --steps 3 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH --output-format csv -d $OUT/sq4 -- $P --mode sync --steps 3 > $OUT/sq4.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $P --mode sync --steps 3 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $P --mode sync --steps 3 > $OUT/write.log 2>&1
if [ "$CFG" = "c5" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf -- $P --mode bf --steps 10 > $OUT/stats_bf.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq_bf -- $P --mode bf --steps 3 > $OUT/sq_bf.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_bf -- $P --mode bf --steps 3 > $OUT/fetch_bf.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_bf -- $P --mode bf --steps 3 > $OUT/write_bf.log 2>&1
fi
python3 tools/prof_config_summary.py $OUT $CFG $R/profiles/${TAG}_${CFG}_pmc.json > $R/profiles/${TAG}_${CFG}_summary.txt
cp $OUT/rate.json $R/profiles/${TAG}_${CFG}_rate.json
cp $OUT/stats/*/*kernel_stats.csv $R/profiles/${TAG}_${CFG}_kernel_stats_sync.csv
cp $OUT/stats_lanes/*/*kernel_stats.csv $R/profiles/${TAG}_${CFG}_kernel_stats_lanes.csv
[ -d $OUT/stats_bf ] && cp $OUT/stats_bf/*/*kernel_stats.csv $R/profiles/${TAG}_${CFG}_kernel_stats_bf_match.csv
mkdir -p $R/gpurun_out/profiles_${TAG}_$CFG && cp $R/profiles/${TAG}_${CFG}_* $R/gpurun_out/profiles_${TAG}_$CFG/
cat $R/profiles/${TAG}_${CFG}_summary.txt
