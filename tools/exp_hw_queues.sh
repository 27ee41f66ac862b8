for q in 4 8 16; do for d in 4 6; do
echo "Q=$q depth=$d: $(GPU_MAX_HW_QUEUES=$q python bench.py --depth $d --no-cpu-baseline --no-single-frame --no-check 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["spread"]["min"], d["spread"]["max"], d["stage_ms_per_step"])')"
done; done
