# k_pyramid_tiles against one launch per level for single frames of several sizes (synchronous device-buffer calls, tools/bench_config.py).
# Recorded with a build whose ORBX_TILES_MAX_PIXELS was not yet capped at 2.5 M pixels (orbx_api.cpp: kPyrTilesMaxPixels): larger frames
# no longer build the tile tables at all.
for cfg in "c3 1" "c3 2" "c5 1" "c2 1" "c2 8"; do set -- $cfg
for px in 0 1000000000; do
echo "$1 batch $2 tiles_max_px=$px: $(ORBX_TILES_MAX_PIXELS=$px python tools/bench_config.py --config $1 --batch $2 --mode sync --steps 50 | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["sync"]["ms_per_batch"],4), d["sync"]["stage_ms"]["pyramid"], d["sync"]["launch"]["pyramid_banded"])')"
done; done
