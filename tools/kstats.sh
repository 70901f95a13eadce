#!/bin/bash
# usage: tools/kstats.sh <tag> <python script + args...>   -- rocprofv3 kernel stats of one command (run on the GPU box from the repo root)
TAG=$1; shift
R=$PWD; O=$R/gpurun_out/ks_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O/run.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("eip::", "").replace("void ", "")
        print("%-40s calls %4s avg_ms %8.4f pct %5.1f" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
tail -2 $O/run.log
