# Wall time of the program's time loop (SLB_TIMING=1) at n-harmonics=128, g-grid=2^20, one GPU: a run with nearly every
# iteration averaged (t-max=0.004: 67 iterations, 63 of them averaged) and one with one in six (t-max=0.3: 363 / 63),
# three steps per launch (default) and two (SLB_FORM=double); exact-order av(), tree av(), no av() (E_omega=0).
O=gpurun_out/prog; mkdir -p $O; cd $O
A="display=4 E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.0038147 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 g-grid=1048576 quiet=1"
for tmax in 0.004 0.3; do
for form in "" double; do
for i in 1 2; do
echo "t-max=$tmax SLB_FORM=${form:-default} run $i"
echo -n "  exact av: "; SLB_FORM=$form SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A t-max=$tmax o=a.txt 2>&1 | grep "time loop" || exit 1
echo -n "  tree av:  "; SLB_FORM=$form SLB_TIMING=1 SLB_AV=tree timeout -k 10 120 ../../bin/boltzmann_solver $A t-max=$tmax o=b.txt 2>&1 | grep "time loop" || exit 1
echo -n "  no av:    "; SLB_FORM=$form SLB_TIMING=1 timeout -k 10 120 ../../bin/boltzmann_solver $A E_omega=0 t-max=$tmax o=c.txt 2>&1 | grep "time loop" || exit 1
done
done
done
