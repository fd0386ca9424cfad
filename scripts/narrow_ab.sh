# Narrow lattices: the double step (four columns per lane) against the general S-steps kernel
# with one column per lane (tuning variants 35..37), over chunk lengths; one GPU.
O=gpurun_out/${1:-narrow}; mkdir -p $O; : > $O/n.txt
timeout -k 10 400 python -u -m pytest tests/test_solver_gpu.py -k multi_step_bit_exact -x -q --timeout 120 --timeout-method thread > $O/test.txt 2>&1 || { tail -20 $O/test.txt; exit 1; }
tail -1 $O/test.txt
for shape in "32 4096 4,8,16,32" "16 20000 4,8,16" "64 262144 8,16,32,64" "128 65536 8,16,32,64,128"; do
  set -- $shape
  echo "== n-harmonics=$1 g-grid=$2: double step (ms per launch of 2 steps)" >> $O/n.txt
  timeout -k 10 200 python scripts/tune.py --double --n-harmonics $1 --g-grid $2 --variants 0 --chunks 0 --blocks 256 --steps 120 2>/dev/null | grep variant >> $O/n.txt || exit 1
  for S in 2 4; do
    echo "== n-harmonics=$1 g-grid=$2: S=$S, one column per lane (ms per launch of $S steps)" >> $O/n.txt
    timeout -k 10 200 python scripts/tune.py --multi $S --n-harmonics $1 --g-grid $2 --variants 35,36 --chunks $3 --blocks 256,64 --steps 120 2>/dev/null | grep variant >> $O/n.txt || exit 1
  done
done
awk '/^==/{print} /variant/{printf "  %s %s %s %s ms  %s\n", $1, $2 $3, $4 $5, $6, $(NF-1)}' $O/n.txt
