# the drop-in call from C++ with the host-side trace of orbx_extract (ORBX_LAT_TRACE).  usage: bash tools/cpp_latency.sh
set -e
R=$GRAFT_REPO_ROOT; cd $R; T=$(mktemp -d)
python3 - <<PY
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_host, orb_slam_tracking_amd as o
from orb_slam_tracking_amd import synth
print(test_host.build_shim_latency(o.lib_path(), "$T"))
a, b = synth.synth_pair(640, 480, 77)
a.tofile("$T/a.raw"); b.tofile("$T/b.raw")
PY
$T/shim_latency 640 480 $T/a.raw $T/b.raw 1000 20 7 300
ORBX_LAT_TRACE=1 $T/shim_latency 640 480 $T/a.raw $T/b.raw 1000 20 7 300 2>&1 | tail -3
