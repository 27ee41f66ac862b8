R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lanes_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 bench.py --steps 60 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame > $OUT/s.log 2>&1
python3 - <<PY
import glob,csv
f=glob.glob("$OUT/s/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if 'orbx' in r['Name']: print("%-50s calls %5s avg %9.1f us  %5.1f %%" % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
