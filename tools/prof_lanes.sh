# kernel trace of the bench workload on the lanes (or --depth 0) and the per-stream timeline of it; usage: bash tools/prof_lanes.sh [depth]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lanes_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 bench.py --depth ${1:-4} --steps 60 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs > $OUT/s.log 2>&1
python3 tools/lane_timeline.py $OUT/s/*/*kernel_trace.csv
