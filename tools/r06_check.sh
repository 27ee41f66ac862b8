#!/bin/bash
# Round-6 loop on the GPU box: the GPU suite, then bench.py, then the real-image bench; everything under gpurun_out/r06/<tag>_*.
# usage: bash tools/r06_check.sh <tag> [pytest args...]
tag=$1; shift
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/r06/${tag}_gputest.log 2>&1; rc=$?
tail -c 1200 gpurun_out/r06/${tag}_gputest.log
[ $rc = 0 ] || exit $rc
python bench.py > gpurun_out/r06/${tag}_bench.json 2> gpurun_out/r06/${tag}_bench.err || { tail -20 gpurun_out/r06/${tag}_bench.err; exit 1; }
python - gpurun_out/r06/${tag}_bench.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print("bench: %.1f k frames/s, %.4f ms/step, checked %s all_checked %s" % (d["value"] / 1e3, d["ms_per_step"], d["checked"], d["all_checked"]))
for k, v in d["other_configs"].items():
    print("  ", k, {a: v[a] for a in v if a in ("frames_per_s_synchronous", "frames_per_s_on_lanes", "stage_ms", "us_per_2000x2000", "checked")})
print("   single_frame", d["single_frame"].get("cpp_shim"))
P
python tools/bench_real_images.py 100 3 > gpurun_out/r06/${tag}_real.log 2>&1; tail -3 gpurun_out/r06/${tag}_real.log
