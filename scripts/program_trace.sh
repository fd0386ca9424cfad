R=$PWD; O=$R/gpurun_out/r03q; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="display=4 E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.0038147 PhiYmin=-4 PhiYmax=4 dt=0.001 n-harmonics=128 g-grid=1048576 quiet=1"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_av -o t -- $R/bin/boltzmann_solver $A t-max=0.004 o=$O/a.txt > $O/trace_av.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_noav -o t -- $R/bin/boltzmann_solver $A E_omega=0 t-max=0.004 o=$O/c.txt > $O/trace_noav.log 2>&1
ls $O/trace_av
