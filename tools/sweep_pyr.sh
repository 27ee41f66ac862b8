# band-count sweep of k_pyramid_bands under the bench default (pipeline depth 3); PYR_KS="2 3 4" bash tools/sweep_pyr.sh
for k in ${PYR_KS:-2 3 4 6}; do
  echo "K=$k: $(ORBX_PYR_BANDS=$k python bench.py --steps 200 --regions 2 --no-cpu-baseline --no-single-frame 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['spread']['median']), round(d['stage_ms_per_step']['pyramid'],4))")"
done
