#!/bin/bash
# The slab program with 2, 4 and 8 processes sharing one GPU (halos staged over gloo) against the
# single-GPU C program on a run that is mostly averaging period: the records must be identical.
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 B=0.0038147 t-max=0.05 quiet=1 omega=10 g-grid=1048576"
timeout -k 10 300 bin/boltzmann_solver $ARGS o=/tmp/spr_c.txt > /dev/null || exit 1
for n in 2 4 8; do
  SLB_DIST_BACKEND=gloo SLB_TIMING=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $((29570 + n)) super-lattice-boltzmann-2d_amd/slab_program.py $ARGS o=/tmp/spr_$n.txt 2>&1 >/dev/null | grep "time loop" || exit 1
  cmp /tmp/spr_c.txt /tmp/spr_$n.txt && echo "$n slabs: record identical to the single-GPU program's" || exit 1
done
tail -1 /tmp/spr_c.txt
