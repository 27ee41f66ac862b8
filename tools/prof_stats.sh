# kernel-stats passes of the bench workload for the current build: every kernel alone on the chip (--depth 0, ORBX_NO_SPLIT=1: one
# 256-frame launch per kernel and step) and live on the four lanes; prints the average launch durations.  usage: bash tools/prof_stats.sh <name>
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
ORBX_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone -- python3 bench.py --depth 0 --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-single-frame --no-check > $OUT/alone.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/live -- python3 bench.py --steps 60 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-check > $OUT/live.log 2>&1
python3 - <<PY
import csv, glob
for tag in ("alone", "live"):
    f = (glob.glob("$OUT/%s/*/*kernel_stats.csv" % tag) + glob.glob("$OUT/%s/*kernel_stats.csv" % tag))[0]
    rows = list(csv.DictReader(open(f)))
    print("== %s (us: avg / min / max, calls)" % tag)
    for r in rows[:8]:
        print("  %-40s %8.1f %8.1f %8.1f %6s" % (r["Name"].split("(")[0][-40:], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Calls"]))
import json
print(open("$OUT/live.log").read().strip().splitlines()[-1][:200])
PY
