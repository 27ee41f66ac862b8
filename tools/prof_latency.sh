set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
python3 tools/latency.py > $OUT/latency.json
cat $OUT/latency.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/latency.py > $OUT/lat_prof.log 2>&1
python3 - <<PY
import glob,csv
f=glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s avg %9.1f ns  total %5.1f %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
