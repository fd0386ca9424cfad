cd /root/repo
ARGS="display=4 E_dc=1 omega=1 mu=5 alpha=1 B=1 PhiYmin=-4 PhiYmax=4 n-harmonics=32 g-grid=4096 dt=0.001 quiet=1 o=/tmp/c1.txt"
for form in double fused sweeps; do
  echo -n "no av, SLB_FORM=$form  "; SLB_FORM=$form SLB_TIMING=1 timeout -k 10 120 bin/boltzmann_solver $ARGS E_omega=0 t-max=10 2>&1 >/dev/null | grep "time loop"
  echo -n "all av, SLB_FORM=$form  "; SLB_FORM=$form SLB_TIMING=1 timeout -k 10 120 bin/boltzmann_solver $ARGS E_omega=0.1 t-max=0.001 2>&1 >/dev/null | grep "time loop"
done
