for v in xa xb xc; do
  export ORBX_LIB=$PWD/orb_slam_tracking_amd/liborbx_$v.so
  bash tools/prof_sel.sh c5 exp_$v 2>&1 | grep -E "k_octree_buckets|k_octree_big|k_octree_global"
done
