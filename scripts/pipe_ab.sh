#!/bin/bash
# The scheduled general kernel (pipe_step_kernel) on one GPU: ms per launch of S steps at
# n-harmonics=128 g-grid=2^20 (and other shapes given as "N M" arguments).
#   70-75 histories in registers, 80-84 in an LDS ring per wave; see slb_step_multi.
out=${OUT:-gpurun_out/pipe_ab.txt}
: > "$out"
shapes=("$@"); [ ${#shapes[@]} -eq 0 ] && shapes=("128 1048576")
for shape in "${shapes[@]}"; do
  set -- $shape
  echo "## n-harmonics=$1 g-grid=$2  double step (fused2_step_kernel), variants 0 and 63" >> "$out"
  timeout -k 10 200 python3 scripts/tune.py --double --n-harmonics $1 --g-grid $2 --steps 48 --variants 0,63,0,63 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
  echo "## S=2 general scheduled kernel" >> "$out"
  timeout -k 10 200 python3 scripts/tune.py --multi 2 --n-harmonics $1 --g-grid $2 --steps 48 --variants 70,72,80,82,84,70,80 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
  echo "## S=3" >> "$out"
  timeout -k 10 300 python3 scripts/tune.py --multi 3 --n-harmonics $1 --g-grid $2 --steps 48 --variants 0,71,73,74,75,81,83,80,82,71,73,81,83 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
