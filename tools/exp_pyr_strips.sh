# Round 5: k_pyramid_bands with column strips for few large frames.  bands_min_frames = 1 forces the banded kernel; K bands x S strips.
# usage (gpurun): bash tools/exp_pyr_strips.sh
OUT=gpurun_out/r05; mkdir -p $OUT
run() { # cfg, env...
  cfg=$1; shift
  echo "$cfg $*: $(env "$@" python tools/bench_config.py --config $cfg --steps 30 --check | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["sync"]["frames_per_s"]), round(d["lanes"]["frames_per_s"]), "pyramid ms", round(d["sync"]["stage_ms"]["pyramid"],4), "banded", d["sync"]["launch"]["pyramid_banded"], d["sync"]["launch"].get("pyramid_bands"), "checked", d.get("checked"))')" | tee -a $OUT/exp_pyr_strips.txt
}
run c5 X=0
for ks in "32 1" "32 2" "32 4" "16 4" "16 8" "24 4" "32 8"; do set -- $ks; run c5 ORBX_BANDS_MIN_FRAMES=1 ORBX_PYR_BANDS=$1 ORBX_PYR_STRIPS=$2; done
run c3 X=0
for ks in "16 1" "16 2" "8 2" "8 4" "12 2"; do set -- $ks; run c3 ORBX_BANDS_MIN_FRAMES=1 ORBX_PYR_BANDS=$1 ORBX_PYR_STRIPS=$2; done
