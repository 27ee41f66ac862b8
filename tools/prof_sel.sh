# kernel stats of one configuration's synchronous call (tools/bench_config.py), selection kernels first.  usage: bash tools/prof_sel.sh c5 tag
set -e
CFG=$1; TAG=$2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04/sel_${TAG}_$CFG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/bench_config.py --config $CFG --mode sync --steps 10 > $OUT/stats.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-60s calls %4s avg %8.1f us  min %8.1f max %8.1f  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
