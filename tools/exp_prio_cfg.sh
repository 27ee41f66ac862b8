# round 4 experiment: s_setprio in the small latency-bound kernels (selection tail, wide-matcher tail) under the two-halves /
# lanes overlap of the large configurations.  usage: bash tools/exp_prio_cfg.sh   (needs liborbx_prio3.so: make VARIANT=prio3 EXTRA=-DORBX_PRIO=3)
for v in "" _prio3 "" _prio3; do
  for c in c5 c3; do
    ORBX_LIB=$PWD/orb_slam_tracking_amd/liborbx$v.so python tools/bench_config.py --config $c --steps 40 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('lib$v', '$c', {k: (round(v['frames_per_s']) if isinstance(v, dict) and 'frames_per_s' in v else None) for k, v in d.items() if k in ('sync','lanes')}, d.get('sync',{}).get('stage_ms'))"
  done
done
