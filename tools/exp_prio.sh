# round 4 experiment: s_setprio at the entry of the latency-bound kernels (ORBX_PRIO build flag), bench.py under each build
for v in "" _prio1 _prio3 "" _prio3 _prio1; do
  export ORBX_LIB=$PWD/orb_slam_tracking_amd/liborbx$v.so
  python bench.py --steps 200 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['value'], d['spread']['in_order'], d['checked'], d['stage_ms_per_step'])"
done
