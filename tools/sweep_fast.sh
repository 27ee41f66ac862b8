cd orb_slam_tracking_amd/csrc
for v in "-DFW_CPW=1 -DFW_XK=8" "-DFW_CPW=1 -DFW_XK=16" "-DFW_CPW=1 -DFW_XK=24" "-DFW_CPW=1 -DFW_XK=32" "-DFW_CPW=1 -DFW_XK=64"; do
  make clean >/dev/null; make EXTRA="$v" >/dev/null 2>&1
  cd ../..; echo "$v: $(ORBX_NO_SPLIT=1 python bench.py --depth 0 --steps 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['stage_ms_per_step']['fast'],4))")"; cd orb_slam_tracking_amd/csrc
done
make clean >/dev/null; make >/dev/null 2>&1
