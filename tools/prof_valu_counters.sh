# which SQ counter measures VALU issue cycles?  runs the issue-rate microbenchmark under rocprofv3 and prints, per opcode
# kernel (n = 8 waves per SIMD launches only), the counters per wave-instruction
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the microbenchmark is always built from its source here (no binary is kept in the tree)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $OUT/valu_rate $R/tools/microbench/valu_rate.hip
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_IOPS SQ_WAVE_CYCLES --output-format csv -d $OUT/mb -- $OUT/valu_rate > $OUT/mb.log 2>&1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/mb/*/*_counter_collection.csv")[0]
rows=list(csv.DictReader(open(f)))
acc=collections.OrderedDict()
for r in rows:
    key=(r['Kernel_Name'][:40], r['Dispatch_Id'])
    acc.setdefault(key, {})[r['Counter_Name']]=float(r['Counter_Value'])
    acc[key]['grid']=int(r['Grid_Size']); acc[key]['lds']=int(r['LDS_Block_Size'])
for (k,d),v in acc.items():
    if v['grid'] != 256*8*4*256 or v.get('SQ_INSTS_VALU',0) < 1e8: continue
    n=v['SQ_INSTS_VALU']
    print(k, ' '.join('%s/inst=%.3f'%(c.replace('SQ_',''), v[c]/n) for c in sorted(v) if c.startswith('SQ_') and c!='SQ_INSTS_VALU'))
PY
