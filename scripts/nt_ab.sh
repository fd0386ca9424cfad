# A/B of the double step's store policy (variant 0 against 27 = non-temporal stores) over the
# benchmark shapes, one GPU.
O=gpurun_out/${1:-nt_ab}; mkdir -p $O
: > $O/nt_ab.txt
for shape in "64 262144" "128 1048576" "128 8388608" "512 1048576" "128 1048577" "32 4194304"; do
  set -- $shape
  echo "== n-harmonics=$1 g-grid=$2" >> $O/nt_ab.txt
  timeout -k 10 200 python scripts/tune.py --double --n-harmonics $1 --g-grid $2 --variants 27,22,27,22 --chunks 0 --blocks 256 --steps 30 2>/dev/null | grep variant >> $O/nt_ab.txt || exit 1
done
cat $O/nt_ab.txt
