# Round 5, second sweep with the final kernel (row pairs, one group per thread): bands x strips of k_pyramid_bands through the knobs
# usage (gpurun): bash tools/exp_pyr_strips2.sh c5|c3
run() { cfg=$1; shift; echo "$cfg $*: $(env "$@" python tools/bench_config.py --config $cfg --steps 30 --check | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["sync"]["frames_per_s"]), round(d["lanes"]["frames_per_s"]), "pyramid ms", round(d["sync"]["stage_ms"]["pyramid"],4), d["sync"]["launch"].get("pyramid_bands"), "checked", d.get("checked"))')"; }
C=${1:-c5}
run $C X=0
if [ $C = c5 ]; then L="32,4 24,4 16,4 16,8 32,2 24,8 12,8"; else L="16,1 16,2 8,4 12,2 8,2 12,4 24,2"; fi
for ks in $L; do run $C ORBX_BANDS_MIN_FRAMES=1 ORBX_PYR_BANDS=${ks%,*} ORBX_PYR_STRIPS=${ks#*,}; done
