set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for form in double fused sweeps; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$form -o $form -- python3 $R/bench.py --no-cpu --form $form --steps 100 --warmup 10 > $O/trace_$form.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${form}_fetch -o f -- python3 $R/bench.py --no-cpu --form $form --steps 10 --warmup 2 > $O/pmc_${form}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${form}_write -o w -- python3 $R/bench.py --no-cpu --form $form --steps 10 --warmup 2 > $O/pmc_${form}_write.log 2>&1
  echo "$form done"
done
timeout -k 10 200 python3 $R/bench.py --form fused > $O/bench_fused.json 2>$O/bench_fused.err
timeout -k 10 200 python3 $R/bench.py --form sweeps > $O/bench_sweeps.json 2>$O/bench_sweeps.err
timeout -k 10 200 python3 $R/bench.py > $O/bench_double.json 2>$O/bench_double.err
ls $O
