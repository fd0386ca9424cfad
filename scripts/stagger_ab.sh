# A/B of the array stagger knob (SLB_STAGGER, bytes between the starts of consecutive lattice
# arrays inside their allocations) on the double step and the two-sweep form, one GPU.
O=gpurun_out/${1:-stagger}; mkdir -p $O
for st in 0 128 256 512 1024 2048 4096 8192 16384 65536 262144 1048576 0; do
  echo "== SLB_STAGGER=$st" >> $O/stagger.txt
  SLB_STAGGER=$st timeout -k 10 120 python scripts/tune.py --double --variants 23 --chunks 0 --blocks 256 --steps 40 2>/dev/null | grep variant >> $O/stagger.txt || exit 1
  SLB_STAGGER=$st timeout -k 10 120 python scripts/tune.py --variants 5 --chunks 0 --blocks 256 --steps 20 2>/dev/null | grep variant >> $O/stagger.txt || exit 1
done
cat $O/stagger.txt
