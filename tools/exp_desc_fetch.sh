# Round 5: where k_describe_patch's time goes.  Builds made before the call:
#   for v in 1 2 3 4; do make -C orb_slam_tracking_amd/csrc VARIANT=dx$v EXTRA=-DORBX_DESC_EXP=$v; done
# 1 = row-per-lane wide loads (real), 2 = tiled reads (timing only), 3 = no window fetch (timing only), 4 = fetch alone (timing only)
# usage (gpurun): bash tools/exp_desc_fetch.sh "" _dx1 _dx2 _dx3 _dx4
OUT=gpurun_out/r05; mkdir -p $OUT
for v in "$@"; do
  L=$PWD/orb_slam_tracking_amd/liborbx$v.so
  a=$(ORBX_LIB=$L ORBX_NO_SPLIT=1 python bench.py --depth 0 --steps 30 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-other-configs --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['stage_ms_per_step']['describe'], round(d['value']))")
  b=$(ORBX_LIB=$L python bench.py --steps 200 --regions 3 --no-cpu-baseline --no-single-frame --no-other-configs --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['stage_ms_per_step']['describe'])")
  echo "lib$v: alone describe ms, single-stream frames/s = $a ; 4 lanes frames/s, live describe ms = $b" | tee -a $OUT/exp_desc_fetch.txt
done
