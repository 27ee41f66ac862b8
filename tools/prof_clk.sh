set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_SPLIT=1
cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- python3 bench.py --depth 0 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame > $OUT/grbm.log 2>&1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/grbm/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m=re.search(r'(k_[a-z_]+)', r['Kernel_Name'])
    if not m: continue
    acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    acc[m.group(1)]['dur'].append((float(r['End_Timestamp'])-float(r['Start_Timestamp'])) if 'End_Timestamp' in r else 0)
for k,v in acc.items():
    g=sum(v['GRBM_GUI_ACTIVE'])/len(v['GRBM_GUI_ACTIVE']); d=sum(v['dur'])/max(len(v['dur']),1)
    print(k, 'GUI_ACTIVE %.4g'%g, 'dur_ns %.0f'%d, 'clock GHz %.2f'%(g/8/d if d else 0))
PY
head -2 $OUT/grbm/*/*_counter_collection.csv | cut -c1-400
