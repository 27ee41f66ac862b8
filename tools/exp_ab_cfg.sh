# A/B of two builds on one box for the large configurations: bash tools/exp_ab_cfg.sh _variant   (liborbx.so against liborbx_variant.so, alternating)
V=$1
for v in "" $V "" $V; do
  for c in c5 c3; do
    ORBX_LIB=$PWD/orb_slam_tracking_amd/liborbx$v.so python tools/bench_config.py --config $c --steps 40 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('lib$v', '$c', 'sync', round(d['sync']['frames_per_s']), 'lanes', round(d['lanes']['frames_per_s']), {k: round(x, 3) for k, x in d['sync']['stage_ms'].items()})"
  done
done
