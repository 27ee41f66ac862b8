# does the step gain when the two issue-bound kernels are capped to a part of a CU each (unused dynamic LDS), so that workgroups of
# two lanes' kernels are co-resident on every CU instead of one kernel filling it?
# k_describe_patch: 17.3 KB per 3-wave workgroup, 9 per CU; pad 9900 -> 6 per CU, 15400 -> 5.  k_fast_wave: 4.5 KB per wave; pad 3700 -> 20 waves, 5700 -> 16
for dp in 0 9900 15400; do for fp in 0 3700 5700; do
echo "descPad=$dp fastPad=$fp: $(ORBX_DESC_LDS_PAD=$dp ORBX_FAST_LDS_PAD=$fp python bench.py --no-cpu-baseline --no-single-frame --no-check --regions 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["stage_ms_per_step"])')"
done; done
