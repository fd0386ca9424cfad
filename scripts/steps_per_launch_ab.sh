#!/bin/bash
# Two steps per launch (fused2_step_kernel) against three (ring_step_kernel) over lattice shapes, one GPU:
# ms per launch and lattice-point-updates per second.  Decides slb_solver_steps_per_launch().
out=${OUT:-gpurun_out/steps_per_launch.txt}
: > "$out"
for shape in "$@"; do
  set -- $shape
  echo "## n-harmonics=$1 g-grid=$2: double step, then three steps per launch" >> "$out"
  timeout -k 10 200 python3 scripts/tune.py --double --n-harmonics $1 --g-grid $2 --steps 48 --variants 0,0 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
  timeout -k 10 200 python3 scripts/tune.py --multi 3 --n-harmonics $1 --g-grid $2 --steps 48 --variants 0,0 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
