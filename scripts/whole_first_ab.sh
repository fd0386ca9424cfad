# Launch schedule of the double step over lattice shapes, one GPU, alternating: 0 = default
# (whole strips first from one round of workgroups up), 34 = the earlier rule (every strip cut
# below four rounds).
O=gpurun_out/${1:-whole}; mkdir -p $O; : > $O/w.txt
for shape in "128 1048576" "512 1048576" "256 1048576" "64 1048576" "128 2097152" "128 3000000" "128 4194304" "128 8388608" "128 1500000" "128 800000" "96 1200000"; do
  set -- $shape
  echo "== n-harmonics=$1 g-grid=$2" >> $O/w.txt
  timeout -k 10 200 python scripts/tune.py --double --n-harmonics $1 --g-grid $2 --variants 34,0,34,0,34,0 --chunks 0 --blocks 256 --steps 30 2>/dev/null | grep variant >> $O/w.txt || exit 1
done
cat $O/w.txt
