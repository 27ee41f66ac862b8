#!/bin/bash
# k_match_bf_mfma without one of its parts (timing-only builds: make VARIANT=bfx<m> EXTRA=-DORBX_BF_EXP=<m>; 1 = no appends,
# 2 = no matrix instructions, 4 = no staging of the next tile): the kernel's average per 64 sets of 2000 x 2000, from rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05/bfx
for v in "$@"; do
  [ "$v" = base ] && s="" || s="_bfx$v"
  out=$R/gpurun_out/r05/bfx/$v
  ORBX_LIB=$R/orb_slam_tracking_amd/liborbx$s.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $R/tools/exp_bf_prof.py 0 > $out.log 2>&1
  python3 - "$out/p_kernel_stats.csv" "$v" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_match_bf_mfma" in r["Name"]: print("ORBX_BF_EXP=%s: k_match_bf_mfma avg %.1f us" % (sys.argv[2], float(r["AverageNs"])/1e3))
P
done
