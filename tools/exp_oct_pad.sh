# selection units per CU against the step: ORBX_OCT_LDS_PAD bytes of unused dynamic LDS per unit (31.2 KB static): 0 -> 5 per CU,
# 9000 -> 4, 22000 -> 3, 49000 -> 2
for pad in 0 9000 22000 49000; do
echo "pad=$pad: $(ORBX_OCT_LDS_PAD=$pad python bench.py --no-cpu-baseline --no-single-frame --no-check 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["stage_ms_per_step"])')"
done
