mkdir -p gpurun_out/c1; cd gpurun_out/c1
A="display=4 E_dc=1 E_omega=0.1 omega=1 mu=5 alpha=1 B=1 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=32 g-grid=4096 t-max=10 quiet=1"
for i in 1 2; do
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A o=a.txt 2>&1 | grep "time loop"
SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A E_omega=0 o=b.txt 2>&1 | grep "time loop"
done
cat a.txt | tail -1 | cut -c1-120
