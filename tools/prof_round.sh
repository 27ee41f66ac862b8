set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_SPLIT=1
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 tools/pmc_summary.py $OUT/sq $OUT/sq2 $OUT/fetch $OUT/write > $OUT/pmc_summary.txt
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("$OUT/stats/*/*kernel_stats.csv")+glob.glob("$OUT/stats/*kernel_stats.csv")
print(open(f[0]).read()[:3000])
PY
