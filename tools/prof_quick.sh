# quick PMC passes (SQ + LDS counters) for the current build; usage: bash tools/prof_quick.sh <name>
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_SPLIT=1
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --depth 0 --steps 10 --warmup 2 --no-cpu-baseline --no-single-frame > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 bench.py --depth 0 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 bench.py --depth 0 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/sq3 -- python3 bench.py --depth 0 --steps 3 --warmup 1 --no-cpu-baseline --no-single-frame > $OUT/sq3.log 2>&1 || true
python3 tools/pmc_summary.py $OUT/sq $OUT/sq2 $OUT/sq3 | grep -E "k_fast|k_describe|k_pyramid|k_octree" > $OUT/pmc_summary.txt
python3 - <<PY
import glob
f=glob.glob("$OUT/stats/*/*kernel_stats.csv")+glob.glob("$OUT/stats/*kernel_stats.csv")
for l in open(f[0]).read().splitlines()[:7]:
    p=l.split('",'); print(p[0][:60], p[1:4])
PY
cat $OUT/pmc_summary.txt
