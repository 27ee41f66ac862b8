#!/bin/bash
# Round-6 fuzz + soak campaign on the GPU box: bash tools/r06_fuzz.sh <tag> <seed0> [scale]
tag=$1; s=$2; k=${3:-1}
mkdir -p gpurun_out/fuzz
for spec in "mixed $((1500*k)) $s" "batched $((200*k)) $((s+1))" "big $((250*k)) $((s+2))" "stateful $((50*k)) $((s+3))"; do
  set -- $spec
  timeout -k 10 700 python tools/fuzz_parity.py $2 $3 $1 > gpurun_out/fuzz/${1}_$tag.log 2>&1
  echo "$1: $(tail -1 gpurun_out/fuzz/${1}_$tag.log)"
done
timeout -k 10 300 python tools/soak.py $((600*k)) > gpurun_out/fuzz/soak_$tag.log 2>&1; echo "soak: $(tail -1 gpurun_out/fuzz/soak_$tag.log)"
