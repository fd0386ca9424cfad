#!/bin/bash
# time per loop iteration of a display=4 run (n-harmonics=128, g-grid=2^20, 629 of 679
# iterations with av()) against the number of streams the exact sums are spread over
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 g-grid=1048576 B=0.0038147 t-max=0.05 quiet=1 o=/tmp/slb_time.txt omega=10"
for q in 4 8; do
  for n in 1 2 3 5 6 7; do
    echo -n "GPU_MAX_HW_QUEUES=$q streams=$n  "
    GPU_MAX_HW_QUEUES=$q SLB_AV_STREAMS=$n SLB_TIMING=1 timeout -k 10 120 bin/boltzmann_solver $ARGS 2>&1 >/dev/null | grep "time loop"
  done
done
