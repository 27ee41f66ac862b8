# A/B of builds on one box, alone-stage times and the four-lane bench: bash tools/exp_ab2.sh <stage> "" _variant ...
# (stage = pyramid | fast | select | describe | match: the stage whose single-stream time is printed)
S=$1; shift
OUT=gpurun_out/r05; mkdir -p $OUT
for v in "$@"; do
  L=$PWD/orb_slam_tracking_amd/liborbx$v.so
  a=$(ORBX_LIB=$L ORBX_NO_SPLIT=1 python bench.py --depth 0 --steps 30 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['stage_ms_per_step']['$S'], round(d['value']))")
  b=$(ORBX_LIB=$L python bench.py --steps 200 --regions 3 --no-cpu-baseline --no-single-frame --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['checked'], d['stage_ms_per_step']['$S'])")
  echo "lib$v: alone $S ms, single-stream frames/s = $a ; 4 lanes frames/s, checked, live $S ms = $b" | tee -a $OUT/exp_ab2.txt
done
