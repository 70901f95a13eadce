#!/bin/bash
# usage: pmc_sq.sh <outdir-tag> <bench args...>   SQ wait/active counters per kernel (one rocprofv3 --pmc pass)
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH SQ_WAVES --output-format csv -d $R/gpurun_out/sq_$TAG -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/sq_$TAG.log 2>&1
cd $R
python - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("eip::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    m={c:max(x) for c,x in v.items()}
    if m.get("SQ_WAVE_CYCLES",0) < 5e6: continue
    wc=m["SQ_WAVE_CYCLES"]
    print("%-34s waves %6d  cycles/wave %9.0f  active %4.1f%% (valu %4.1f%%)  wait_any %4.1f%%  wait_inst %4.1f%%  valu/wave %8.0f  ifetch/wave %7.0f" % (k[:34], m["SQ_WAVES"], wc/m["SQ_WAVES"], 100*m["SQ_ACTIVE_INST_ANY"]/wc, 100*m["SQ_ACTIVE_INST_VALU"]/wc, 100*m["SQ_WAIT_ANY"]/wc, 100*m["SQ_WAIT_INST_ANY"]/wc, m["SQ_INSTS_VALU"]/m["SQ_WAVES"], m.get("SQ_IFETCH",0)/m["SQ_WAVES"]))
PY
