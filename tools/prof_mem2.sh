# second set of memory-side PMC passes for one kernel: L1->L2 latency, L2 hit / miss, L2->fabric requests
# usage: bash tools/prof_mem2.sh <name> <kernel regex>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ORBX_NO_SPLIT=1
cd $R
P="python3 bench.py --depth 0 --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-single-frame"
pass() { n=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $P > $OUT/$n.log 2>&1; echo "pass $n rc=$?" | tee -a $OUT/progress.txt; }
pass l1 TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_ACCESSES
pass l2a TCC_HIT TCC_MISS TCC_READ TCC_WRITE
pass l2b TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B
pass l2c TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY
pass sq SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $OUT/l1 $OUT/l2a $OUT/l2b $OUT/l2c $OUT/sq | grep -E "$2" | tr ' ' '\n' > $OUT/summary.txt
cat $OUT/summary.txt
