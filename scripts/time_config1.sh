#!/bin/bash
# wall time of the reference's own CPU-runnable case (n-harmonics=32, g-grid=4096, 16281
# iterations, the last 6284 with av()) in the program's step forms
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 omega=1 mu=5 alpha=1 B=1 PhiYmin=-4 PhiYmax=4 n-harmonics=32 g-grid=4096 dt=0.001 t-max=10 quiet=1 o=/tmp/c1.txt"
for form in double fused sweeps; do
  for avm in exact tree; do
    echo -n "SLB_FORM=$form SLB_AV=$avm  "
    SLB_FORM=$form SLB_AV=$avm SLB_TIMING=1 timeout -k 10 120 bin/boltzmann_solver $ARGS 2>&1 >/dev/null | grep "time loop"
  done
done
tail -1 /tmp/c1.txt | cut -d' ' -f5-8
