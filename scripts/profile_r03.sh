# rocprofv3 kernel-trace statistics and HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass each) of
# `python3 bench.py` on one GPU: three steps per launch (the default; ring_step_kernel) and two (fused2_step_kernel)
# at the three full lattice shapes, the two-sweep form at n-harmonics=128, g-grid=2^20, and config 2's shape.
# Results under gpurun_out/$1.
set -e
R=$PWD; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {   # tag, bench arguments...
  tag=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o t -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 "$@" > $O/trace_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$tag -o f -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 --steps 12 --warmup 2 "$@" > $O/fetch_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$tag -o w -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 --steps 12 --warmup 2 "$@" > $O/write_$tag.log 2>&1
  echo "$tag done"
}
run triple_n128_g1048576 --steps 99 --warmup 6
run double_n128_g1048576 --form double --steps 100 --warmup 10
run sweeps_n128_g1048576 --form sweeps --steps 50 --warmup 6
run triple_n512_g1048576 --n-harmonics 512 --steps 30 --warmup 6
run double_n512_g1048576 --form double --n-harmonics 512 --steps 30 --warmup 4
run triple_n128_g8388608 --g-grid-per-gpu 8388608 --steps 18 --warmup 6
run double_n128_g8388608 --form double --g-grid-per-gpu 8388608 --steps 20 --warmup 4
run double_n64_g262144 --form double --n-harmonics 64 --g-grid-per-gpu 262144 --steps 100 --warmup 10
cd $R
for t in triple_n128_g1048576 double_n128_g1048576 sweeps_n128_g1048576 triple_n512_g1048576 double_n512_g1048576 triple_n128_g8388608 double_n128_g8388608 double_n64_g262144; do
  tail -1 $O/trace_$t.log | cut -c1-300
done
