# Wall time of the program's time loop (SLB_TIMING=1) at n-harmonics=128, g-grid=2^20 with nearly every
# iteration averaged: exact-order av(), tree av(), no av() (E_omega=0); one GPU.
O=gpurun_out/prog; mkdir -p $O; cd $O
A="display=4 E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.0038147 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 g-grid=1048576 quiet=1"
for i in 1 2; do
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A t-max=0.004 o=a.txt 2>&1 | grep "time loop"
SLB_TIMING=1 SLB_AV=tree timeout -k 10 120 ../../bin/boltzmann_solver $A t-max=0.004 o=b.txt 2>&1 | grep "time loop"
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A E_omega=0 t-max=0.004 o=c.txt 2>&1 | grep "time loop"
done
