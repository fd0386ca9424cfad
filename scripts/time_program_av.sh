#!/bin/bash
# Wall time of the averaging period of a display=4 run at n-harmonics=128, g-grid=2^20:
# 628 loop iterations with av() after 50 without (omega=10, t-max=0.05), against the same
# run with 63 averaging iterations (omega=100) to take the start-up out.
set -e
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 g-grid=1048576 B=0.0038147 t-max=0.05 quiet=1 o=/tmp/slb_time.txt"
for mode in exact chain tree; do
  for omega in 100 10; do
    s=$(date +%s%N)
    SLB_TIMING=1 SLB_AV=$mode timeout -k 10 300 bin/boltzmann_solver $ARGS omega=$omega > /dev/null
    e=$(date +%s%N)
    echo "SLB_AV=$mode omega=$omega: $(( (e - s) / 1000000 )) ms   $(tail -1 /tmp/slb_time.txt | cut -d' ' -f5,6,10)"
  done
done
