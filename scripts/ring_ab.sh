#!/bin/bash
# ring_step_kernel (tuning variants 90, 91 of slb_step_multi: the LDS-ring kernel with branch-free
# steady passes and the next pass's rows requested a pass ahead) against pipe_step_kernel's ring form
# (80, 81) and the double step, on one GPU: ms per launch of S steps.  Shapes as "N M" arguments.
out=${OUT:-gpurun_out/ring_ab.txt}
: > "$out"
shapes=("$@"); [ ${#shapes[@]} -eq 0 ] && shapes=("128 1048576")
for shape in "${shapes[@]}"; do
  set -- $shape
  echo "## n-harmonics=$1 g-grid=$2  double step (fused2_step_kernel), variants 0 and 63" >> "$out"
  timeout -k 10 200 python3 scripts/tune.py --double --n-harmonics $1 --g-grid $2 --steps 48 --variants 0,63,0,63 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
  echo "## S=2: 80 pipe ring, 90 / 91 ring_step_kernel at 2 / 3 waves per SIMD" >> "$out"
  timeout -k 10 200 python3 scripts/tune.py --multi 2 --n-harmonics $1 --g-grid $2 --steps 48 --variants 80,90,91,80,90,91 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
  echo "## S=3: 81 pipe ring, 90 ring_step_kernel" >> "$out"
  timeout -k 10 300 python3 scripts/tune.py --multi 3 --n-harmonics $1 --g-grid $2 --steps 48 --variants 81,90,81,90 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
