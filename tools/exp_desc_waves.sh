cd orb_slam_tracking_amd/csrc
for w in 2 4 3; do
sed -i "s/^#define DESC_WAVES [0-9] /#define DESC_WAVES $w /" orbx_kernels.hip
make 2>&1 | grep -E "error"
cd ../..; ORBX_NO_SPLIT=1 python bench.py --depth 0 --steps 30 --warmup 3 --regions 1 --no-cpu-baseline --no-single-frame --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('DESC_WAVES=$w alone-ish describe ms', d['stage_ms_per_step']['describe'])"
python bench.py --no-cpu-baseline --no-single-frame --no-check --regions 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   bench', round(d['value']))"
cd orb_slam_tracking_amd/csrc
done
