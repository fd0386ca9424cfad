#!/bin/bash
# kernel statistics of the slab program's averaging period (one process, exact sums)
cd "$(dirname "$0")/.."
ARGS="display=4 E_dc=1 E_omega=0.1 mu=5 alpha=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 B=0.0038147 t-max=0.05 quiet=1 omega=10 g-grid=1048576 o=/tmp/sp_prof.txt"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_slab_av
SLB_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_slab_av -- python3 super-lattice-boltzmann-2d_amd/slab_program.py $ARGS
