#!/bin/bash
# The display=77 trace at n-harmonics=64, g-grid=262144 (the second named configuration) by the single-GPU C program
# and by the slab program with 4 processes sharing the GPU (gloo): the traces must be identical.
cd "$(dirname "$0")/.."
ARGS="display=77 E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.015 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=64 g-grid=262144 t-max=0.05 quiet=1"
timeout -k 10 300 bin/boltzmann_solver $ARGS o=/tmp/c2_c.txt > /dev/null || exit 1
SLB_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29588 super-lattice-boltzmann-2d_amd/slab_program.py $ARGS o=/tmp/c2_4.txt > /dev/null 2>&1 || exit 1
cmp /tmp/c2_c.txt /tmp/c2_4.txt && echo "4 slabs: trace identical to the single-GPU program's ($(grep -vc '^#' /tmp/c2_c.txt) records)" || exit 1
grep -v '^#' /tmp/c2_c.txt | cut -d' ' -f5,7,14
