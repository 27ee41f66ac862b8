# memory-pipeline PMC passes (TA / TCP / TD / SQ VMEM counters) for one kernel of the current build
# usage: bash tools/prof_mem.sh <name> <kernel regex>     (gpurun; separate passes, --kernel-trace only; at most two
# counters per TA / TD block and pass: more is refused with "exceeds the capabilities of the hardware")
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_SPLIT=1
cd $R
P="python3 bench.py --depth 0 --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs"
pass() { n=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $P > $OUT/$n.log 2>&1; echo "pass $n rc=$?" | tee -a $OUT/progress.txt; }
pass ta1 TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TD_TD_BUSY TD_TC_STALL TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
pass tcp TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1 TCP_TOTAL_READ
pass sqa SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass sqb SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVES SQ_INSTS_VALU
python3 tools/pmc_summary.py $OUT/ta1 $OUT/ta2 $OUT/tcp $OUT/sqa $OUT/sqb | grep -E "$2" | tr ' ' '\n' > $OUT/summary.txt
cat $OUT/summary.txt
