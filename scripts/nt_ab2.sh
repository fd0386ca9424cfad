# store policy of the two-sweep form (4 plain, 5 non-temporal loads and stores, 9 stores only)
# and of the one-launch step (6 plain, 7 both, 8 stores only), one GPU
O=gpurun_out/${1:-nt_ab}; mkdir -p $O
: > $O/nt_ab2.txt
for shape in "64 262144" "128 1048576" "512 1048576"; do
  set -- $shape
  echo "== two sweeps n-harmonics=$1 g-grid=$2" >> $O/nt_ab2.txt
  timeout -k 10 200 python scripts/tune.py --n-harmonics $1 --g-grid $2 --variants 4,5,9,4,5,9 --chunks 0 --blocks 256 --steps 20 2>/dev/null | grep variant >> $O/nt_ab2.txt || exit 1
  echo "== one-launch step n-harmonics=$1 g-grid=$2" >> $O/nt_ab2.txt
  timeout -k 10 200 python scripts/tune.py --fused --n-harmonics $1 --g-grid $2 --variants 6,7,8,6,7,8 --chunks 0 --blocks 256 --steps 20 2>/dev/null | grep variant >> $O/nt_ab2.txt || exit 1
done
cat $O/nt_ab2.txt
