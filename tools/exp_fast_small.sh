# k_fast (a workgroup of four waves per FAST cell) against k_fast_wave (one wave per cell) for small batches: synchronous calls,
# frames resident (tools/batch_sweep.py); ORBX_FAST_WG_MAX_CELLS = cells per launch up to which k_fast is taken (577 per 640x480 frame)
for mc in 0 1024 2500 5000; do
echo "max_cells=$mc: $(ORBX_FAST_WG_MAX_CELLS=$mc SWEEP_B=1,2,4,8 python tools/batch_sweep.py | python -c '
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["batch"], d["sync"]["ms_per_call"], round(d["lanes"]["frames_per_s"]), end=" | ")')"
done
