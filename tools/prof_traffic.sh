# HBM-side bytes per frame (rocprofv3 FETCH_SIZE and WRITE_SIZE, separate passes) of the whole hot path at 1, 32 and 256 frames
# per call; usage: bash tools/prof_traffic.sh <name>     -> gpurun_out/<name>/traffic.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for B in 1 32 256; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$B -- python3 tools/traffic_run.py $B 4 > $OUT/f$B.log 2>&1; echo "fetch B=$B rc=$?" >> $OUT/progress.txt
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$B -- python3 tools/traffic_run.py $B 4 > $OUT/w$B.log 2>&1; echo "write B=$B rc=$?" >> $OUT/progress.txt
done
python3 - <<PY > $OUT/traffic.txt
import csv, glob, collections
for B in (1, 32, 256):
    tot = {}
    per = collections.defaultdict(lambda: [0.0, 0.0])
    for i, c in enumerate(("f", "w")):
        fs = glob.glob("$OUT/%s%d/*/*_counter_collection.csv" % (c, B))
        s = 0.0
        for r in csv.DictReader(open(fs[0])):
            kn = r["Kernel_Name"]
            if "orbx" not in kn: continue
            v = float(r["Counter_Value"]) * 1024.0   # KiB units
            s += v
            import re
            m = re.search(r"(k_[a-z_]+)", kn)
            per[m.group(1) if m else kn[:20]][i] += v
        tot[c] = s
    calls = 4
    print("B=%d: fetch %.0f B, write %.0f B per frame (all orbx kernels, %d calls)" % (B, tot["f"] / (B * calls), tot["w"] / (B * calls), calls))
    for kname, (f, w) in sorted(per.items(), key=lambda x: -x[1][0] - x[1][1]):
        print("    %-24s fetch %10.0f  write %10.0f  B per frame" % (kname, f / (B * calls), w / (B * calls)))
PY
cat $OUT/traffic.txt
