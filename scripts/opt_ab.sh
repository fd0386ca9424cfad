#!/bin/bash
# A/B of the double step's point-update forms on one GPU (tune.py --double, ms per launch of two steps):
# 0 = the compiler's own (default before), 60 = row factors under a uniform branch + dt*a0 carried,
# 62 = 60 spelled two columns wide, 63 = 62 with the shared-reciprocal division, 61 = 60 + that division
out=${1:-gpurun_out/opt_ab.txt}
: > "$out"
for shape in "128 1048576" "512 1048576" "128 8388608" "64 262144"; do
  set -- $shape
  echo "## n-harmonics=$1 g-grid=$2" >> "$out"
  timeout -k 10 300 python3 scripts/tune.py --double --n-harmonics $1 --g-grid $2 --steps 40 \
      --variants 0,60,62,63,61,0,60,62,63,61 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
