#!/bin/bash
# A/B of one diagnostic knob on one box: alone stage time (one stream) and the four-lane bench, alternating.
# usage: bash tools/exp_knob_ab.sh <stage> <ORBX_KNOB_NAME> [value]      e.g. fast ORBX_FAST_ONE_LAUNCH 1
S=$1; K=$2; V=${3:-1}
mkdir -p gpurun_out/r06
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $K=$V; else unset $K; fi
    a=$(ORBX_NO_SPLIT=1 python bench.py --depth 0 --steps 30 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['stage_ms_per_step']['$S'], round(d['value']))")
    b=$(python bench.py --steps 200 --regions 3 --no-cpu-baseline --no-single-frame --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['checked'], d['stage_ms_per_step']['$S'])")
    echo "$K=$([ $on = 1 ] && echo $V || echo unset): alone $S ms, single-stream frames/s = $a ; 4 lanes frames/s, checked, live $S ms = $b" | tee -a gpurun_out/r06/exp_knob_ab.txt
  done
done
