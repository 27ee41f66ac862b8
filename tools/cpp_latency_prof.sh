# kernel trace of the C++ drop-in call (tests/cpp/shim_latency.cpp): per-kernel durations and the gaps between the kernels of one frame
set -e
R=$GRAFT_REPO_ROOT; cd $R; T=$(mktemp -d); OUT=$R/gpurun_out/r04/cpp_lat; mkdir -p $OUT
python3 - <<PY
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_host, orb_slam_tracking_amd as o
from orb_slam_tracking_amd import synth
test_host.build_shim_latency(o.lib_path(), "$T")
a, b = synth.synth_pair(640, 480, 77)
a.tofile("$T/a.raw"); b.tofile("$T/b.raw")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -- $T/shim_latency 640 480 $T/a.raw $T/b.raw 1000 20 7 100 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/trace/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames of the timed loop: a frame = k_pyramid_tiles ... k_describe_patch
names = [r["Kernel_Name"].split("(")[0].replace("void orbx::", "").replace("orbx::", "")[:28] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("k_pyramid_tiles")]
import statistics
per = {}
gaps = []
spans = []
for s0, s1 in zip(starts[60:160], starts[61:161]):
    fr = rows[s0:s1]
    if not names[s1 - 1].startswith("k_describe"):
        continue
    spans.append((int(fr[-1]["End_Timestamp"]) - int(fr[0]["Start_Timestamp"])) / 1e3)
    for i, r in enumerate(fr):
        n = names[s0 + i]
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if i:
            gaps.append((int(r["Start_Timestamp"]) - int(fr[i - 1]["End_Timestamp"])) / 1e3)
    gaps_frame = (int(rows[s1]["Start_Timestamp"]) - int(fr[-1]["End_Timestamp"])) / 1e3
    per.setdefault("(gap to the next frame's first kernel)", []).append(gaps_frame)
for n, v in per.items():
    print("%-40s median %7.1f us" % (n, statistics.median(v)))
print("first kernel start -> last kernel end: median %.1f us; gaps between a frame's kernels: median %.1f us each, sum %.1f" % (
    statistics.median(spans), statistics.median(gaps), statistics.median(gaps) * 4))
PY
