# rocprofv3 kernel-trace statistics and HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass
# each) of `python3 bench.py` for the three full lattice shapes on one GPU, double step; and for
# the two-sweep form at n-harmonics=128, g-grid=2^20.  Results under gpurun_out/$1.
set -e
R=$PWD; O=$R/gpurun_out/${1:-r02p}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {   # tag, bench arguments...
  tag=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -o t -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 "$@" > $O/trace_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$tag -o f -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 --steps 8 --warmup 2 "$@" > $O/fetch_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$tag -o w -- python3 $R/bench.py --no-cpu --no-two-sweep --repeats 1 --steps 8 --warmup 2 "$@" > $O/write_$tag.log 2>&1
  echo "$tag done"
}
run double_n128_g1048576 --steps 100 --warmup 10
run sweeps_n128_g1048576 --form sweeps --steps 50 --warmup 6
run fused_n128_g1048576 --form fused --steps 50 --warmup 6
run double_n512_g1048576 --n-harmonics 512 --steps 30 --warmup 4
run double_n512_g1048576_chunk256 --n-harmonics 512 --n-chunk 256 --steps 30 --warmup 4
run double_n128_g8388608 --g-grid-per-gpu 8388608 --steps 20 --warmup 4
run double_n128_g8388608_chunk64 --g-grid-per-gpu 8388608 --n-chunk 64 --steps 20 --warmup 4
cd $R
for t in double_n128_g1048576 sweeps_n128_g1048576 fused_n128_g1048576 double_n512_g1048576 double_n512_g1048576_chunk256 double_n128_g8388608 double_n128_g8388608_chunk64; do
  tail -1 $O/trace_$t.log | cut -c1-400
done
