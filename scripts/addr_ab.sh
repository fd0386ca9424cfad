#!/bin/bash
# A/B of the double step's addressing forms (libraries built with -DSLB_ADDR=0/1/2 under ab/) and of
# a0 loaded (52) / formed (0, 51) / formed in single precision as a timing probe (53).
out=${1:-gpurun_out/addr_ab.txt}
: > "$out"
for a in 0 1 2 0 1 2; do
  echo "## SLB_ADDR=$a  n-harmonics=128 g-grid=1048576" >> "$out"
  SLB_LIB=$PWD/ab/libslb2d_addr$a.so timeout -k 10 200 python3 scripts/tune.py --double --steps 40 \
      --variants 52,0,51,53,52,0,51,53 --chunks 0 --blocks 256 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
