#!/bin/bash
# SQ counters of the S-steps-per-launch kernels on one GPU (one rocprofv3 pass per tuning variant):
# where the waves' cycles go -- issuing, parked at a wait, stalled at issue.
#   usage: scripts/pmc_sq.sh OUTDIR S "variants" [N M]
set -e
R=$PWD; O=$R/gpurun_out/$1; S=$2; VARS=$3; N=${4:-128}; M=${5:-1048576}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in $VARS; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d $O/sq_S${S}_v$v -o c -- python3 $R/scripts/tune.py --multi $S --n-harmonics $N --g-grid $M --steps 12 --variants $v --chunks 0 --blocks 256 > $O/sq_S${S}_v$v.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU \
    --output-format csv -d $O/sq2_S${S}_v$v -o c -- python3 $R/scripts/tune.py --multi $S --n-harmonics $N --g-grid $M --steps 12 --variants $v --chunks 0 --blocks 256 > $O/sq2_S${S}_v$v.log 2>&1
done
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(O + "/sq*_S*_v*/")):
    acc = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "step_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    line = d.rstrip("/").split("/")[-1] + ": " + "  ".join(f"{k}={sum(v[-12:])/len(v[-12:]):.4g}" for k, v in sorted(acc.items()))
    print(line)
PY
