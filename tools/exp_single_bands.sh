# single-frame latency with the one-launch banded pyramid forced on (ORBX_BANDS_MIN_FRAMES=1) and K thin bands
echo "per-level launches: $(python tools/latency.py)"
for k in 8 16 24 32; do echo "bands=$k: $(ORBX_BANDS_MIN_FRAMES=1 ORBX_PYR_BANDS=$k python tools/latency.py)"; done
