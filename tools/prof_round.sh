# One round of committed profiles (run on the MI355X box through gpurun): rocprofv3 kernel stats of the default bench command
# and of the single-stream variant, PMC passes (SQ counters, VALU issue counters, FETCH_SIZE, WRITE_SIZE: separate passes, as
# the guide prescribes), summarised into profiles/<tag>_*.  usage: bash tools/prof_round.sh r02_a
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 bench.py --steps 100 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs > $OUT/stats2.log 2>&1
export ORBX_NO_SPLIT=1   # with --depth 0: one stream, one 256-frame launch per kernel and step (every kernel alone on the chip)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --depth 0 --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs > $OUT/stats.log 2>&1
P="python3 bench.py --depth 0 --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq -- $P > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH --output-format csv -d $OUT/sq4 -- $P > $OUT/sq4.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $P > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $P > $OUT/write.log 2>&1
python3 tools/pmc_to_json.py $OUT 256 $R/profiles/${TAG}_pmc.json "rocprofv3 --kernel-trace --pmc <SQ counters | SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 ... | FETCH_SIZE | WRITE_SIZE> (separate passes) -- python3 bench.py --depth 0 --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs with ORBX_NO_SPLIT=1 (one 256-frame launch per kernel and step); durations from the --stats pass of the same single-stream command" > $OUT/pmc_table.txt
cp $OUT/pmc_table.txt $R/profiles/${TAG}_pmc_table.txt
python3 tools/pmc_summary.py $OUT/sq $OUT/sq4 $OUT/fetch $OUT/write > $R/profiles/${TAG}_pmc_counters.txt
cp $OUT/stats/*/*kernel_stats.csv $R/profiles/${TAG}_kernel_stats_single_stream.csv
cp $OUT/stats2/*/*kernel_stats.csv $R/profiles/${TAG}_kernel_stats.csv
# the bench line last: its roofline object reads the newest profiles/r*_pmc.json, i.e. the counters collected just above
unset ORBX_NO_SPLIT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/bench.json $R/profiles/${TAG}_bench.json
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/
cat $OUT/pmc_table.txt
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['spread'], d['roofline']['kernel'], d['roofline'].get('frac'), d.get('cpu_baseline',{}).get('value'))"
