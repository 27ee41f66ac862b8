#!/bin/bash
# k_fast_wave alone (one lane: nothing overlaps) on real and synthetic frames, per library variant: rocprofv3 kernel averages.
# usage (on the GPU box): bash tools/exp_fast_flat.sh <variant> ...     variants = suffixes of orb_slam_tracking_amd/liborbx<suffix>.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05/flat
for v in "$@"; do
  [ "$v" = base ] && s="" || s="_$v"
  for data in real synth; do
    [ $data = synth ] && extra="x" || extra=""
    out=$R/gpurun_out/r05/flat/${v}_$data
    ORBX_LIB=$R/orb_slam_tracking_amd/liborbx$s.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $R/tools/bench_real_images.py 30 1 1 $extra > $out.log 2>&1
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "$v $data: $(grep "frames/s" $out.log | cut -c1-90)"
    python3 - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("k_fast_wave","k_describe","k_pyramid")): print("   %-40s calls %s avg %.1f us" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3))
P
  done
done
