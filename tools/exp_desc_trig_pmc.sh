cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export ORBX_NO_SPLIT=1
for v in "" _notrig; do
  export ORBX_LIB=$R/orb_slam_tracking_amd/liborbx$v.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r05/pd$v -o p -- python3 bench.py --depth 0 --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs > /dev/null 2>&1
  python3 - $R/gpurun_out/r05/pd$v "$v" <<'P'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+"/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if "k_describe" in r["Kernel_Name"]:
        acc[r["Counter_Name"]][0]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("lib%s:" % sys.argv[2], {k: round(v[0]/n[k]/256) for k,v in acc.items()}, "per frame")
P
done
