# A/B of two builds on one box: bash tools/exp_ab.sh _variant [bench.py flags]   (liborbx.so against liborbx_variant.so, alternating)
V=$1; shift
for v in "" $V "" $V "" $V; do
  ORBX_LIB=$PWD/orb_slam_tracking_amd/liborbx$v.so python bench.py --steps 200 --no-cpu-baseline --no-single-frame --no-other-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lib$v', round(d['value']), d['checked'], d['spread']['in_order'])"
done
