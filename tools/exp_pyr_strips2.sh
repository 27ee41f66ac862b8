run() { cfg=$1; shift; echo "$cfg $*: $(env "$@" python tools/bench_config.py --config $cfg --steps 30 --check | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["sync"]["frames_per_s"]), round(d["lanes"]["frames_per_s"]), "pyramid ms", round(d["sync"]["stage_ms"]["pyramid"],4), d["sync"]["launch"].get("pyramid_bands"), "checked", d.get("checked"))')"; }
run c5 X=0
for ks in "32 4" "24 4" "16 4" "16 8" "32 2" "24 8" "12 8"; do set -- $ks; run c5 ORBX_BANDS_MIN_FRAMES=1 ORBX_PYR_BANDS=$1 ORBX_PYR_STRIPS=$2; done
