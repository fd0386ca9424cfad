# Two builds of the library on one GPU, alternating: ms per launch of three steps at n-harmonics=128, g-grid=2^20
# (and of two steps), each build several times.  usage: bash scripts/lib_ab.sh libA.so libB.so
A=$1; B=$2
for i in 1 2 3 4; do
  for L in $A $B; do
    echo -n "$(basename $L)  "
    SLB_LIB=$L timeout -k 10 100 python3 scripts/tune.py --multi 3 --variants 0 --chunks 0 --blocks 256 --steps 240 2>/dev/null | grep variant | sed 's/variant=0 n_chunk=   0 block= 256//'
  done
done
