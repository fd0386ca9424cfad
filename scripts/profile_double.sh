# rocprofv3 passes for the default (double step) form: kernel trace + stats, then the two HBM
# counters in passes of their own; results under gpurun_out/r01c
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_double -o double -- python3 $R/bench.py --no-cpu --steps 100 --warmup 10 > $O/trace_double.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_double_fetch -o f -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/pmc_double_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_double_write -o w -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/pmc_double_write.log 2>&1
python3 $R/scripts/pmc_summary.py $O/pmc_double_fetch $O/pmc_double_write fused2_step_kernel > $O/double_pmc_hbm.json
find $O/trace_double -name "*kernel_stats.csv" -exec cp {} $O/double_kernel_stats.csv \;
timeout -k 10 200 python3 $R/bench.py > $O/bench_double.json 2>$O/bench_double.err
ls $O
