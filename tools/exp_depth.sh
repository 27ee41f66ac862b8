for d in 0 2 3 4 6; do
echo "depth=$d: $(python bench.py --depth $d --no-cpu-baseline --no-single-frame --no-check --regions 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]), d["spread"]["min"], d["spread"]["max"])')"
done
