# Wall time of the time loop (SLB_TIMING=1) of the display=77 trace at n-harmonics=64,
# g-grid=2^18 (113 iterations, 11 records), and of the same lattice with display=4; one GPU.
mkdir -p gpurun_out/c2; cd gpurun_out/c2
A="E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.015 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=64 g-grid=262144 t-max=0.05 quiet=1"
for i in 1 2; do
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver display=77 $A o=a.txt 2>&1 | grep "time loop"
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver display=4 $A o=b.txt 2>&1 | grep "time loop"
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver display=4 $A E_omega=0 o=c.txt 2>&1 | grep "time loop"
done
