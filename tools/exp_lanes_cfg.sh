# round 4 experiment: lanes (pipeline depth) and hardware queues for the large configurations
for q in 4 8; do for d in 2 3 4 6; do for c in c5 c3; do
  GPU_MAX_HW_QUEUES=$q python tools/bench_config.py --config $c --steps 40 --depth $d --mode lanes 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('queues $q depth $d $c', round(d['lanes']['frames_per_s']))"
done; done; done
