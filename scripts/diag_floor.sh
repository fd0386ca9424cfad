# Memory floor of the double step (arithmetic removed, tuning variant 98 of a library built
# with -DSLB_TIMING_VARIANTS) next to the real kernel, for the launch schedules; then SQ
# counters and the busy-cycle count of the real kernel (clock = GRBM_GUI_ACTIVE / duration).
O=$PWD/gpurun_out/${1:-diag}; mkdir -p $O; R=$PWD
SLB_LIB=$R/ab/libslb2d_diag.so timeout -k 10 200 python scripts/tune.py --double --variants 0,98,0,98,0,98 --chunks 0 --blocks 256 --steps 40 > $O/floor.txt 2>&1 || exit 1
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py --no-cpu --no-two-sweep --steps 10 --warmup 2 --repeats 1 > $O/pmc_sq.log 2>&1 || exit 1
cd $R; cat $O/floor.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("$O/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in acc.items(): print(k, sum(v)/len(v))
if dur: print("duration_ns", sum(dur)/len(dur))
PY
