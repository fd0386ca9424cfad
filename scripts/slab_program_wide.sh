#!/bin/bash
# Two slabs of 2^20 columns each (the per-GPU width of the scaling bench) sharing one GPU over
# gloo, against the single-GPU C program at g-grid=2^21: the display=4 records must be identical.
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 B=0.0019073 t-max=0.02 quiet=1 omega=100 g-grid=2097152"
timeout -k 10 300 bin/boltzmann_solver $ARGS o=/tmp/spw_c.txt > /dev/null || exit 1
SLB_DIST_BACKEND=gloo SLB_TIMING=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29591 super-lattice-boltzmann-2d_amd/slab_program.py $ARGS o=/tmp/spw_2.txt 2>&1 >/dev/null | grep "time loop" || exit 1
cmp /tmp/spw_c.txt /tmp/spw_2.txt && echo "2 slabs of 2^20 columns: record identical to the single-GPU program's" || exit 1
tail -1 /tmp/spw_c.txt
