#!/bin/bash
# A/B on one GPU: the double step with a0 formed from its factors (variant 0, the default when
# the engine's check passed; 51: formed per stage) against a0 loaded (variant 52), alternating
# inside one process per shape.  usage: scripts/a0_ab.sh [out-file]
out=${1:-gpurun_out/a0_ab.txt}
: > "$out"
for shape in "128 1048576" "512 1048576" "128 8388608" "64 262144"; do
  set -- $shape
  echo "## n-harmonics=$1 g-grid=$2 (ms per launch of two steps; GB/s on 36 B per cell)" >> "$out"
  timeout -k 10 300 python3 scripts/tune.py --double --n-harmonics $1 --g-grid $2 --steps 40 \
      --variants 52,0,51,52,0,51,52,0,51 --chunks 0 --blocks 256 >> "$out" 2>&1 || exit 1
done
