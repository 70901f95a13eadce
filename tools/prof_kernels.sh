#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <bench.py args...>     (run on the GPU box from the repo root)
# Four rocprofv3 passes of the same bench.py command -- kernel trace + stats, then FETCH_SIZE, WRITE_SIZE
# and SQ counters each in their own --pmc pass (they do not fit one pass; MI355X_MICROARCH.md "PMC slots")
# -- summarised per kernel into gpurun_out/<tag>_kernels.csv (copy into profiles/ to commit).
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-host-abi --sustained 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py "$@" $COMMON > $O/ks.log 2>&1 || echo "ks pass failed" >> $O/ks.log
echo "ks done" >> $O/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" $COMMON --steps 2 --warmup 1 > $O/fetch.log 2>&1 || echo "fetch pass failed" >> $O/fetch.log
echo "fetch done" >> $O/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" $COMMON --steps 2 --warmup 1 > $O/write.log 2>&1 || echo "write pass failed" >> $O/write.log
echo "write done" >> $O/progress.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -- python3 $R/bench.py "$@" $COMMON --steps 2 --warmup 1 > $O/sq.log 2>&1 || echo "sq pass failed" >> $O/sq.log
echo "sq done" >> $O/progress.log
cd $R
python3 - "$O" "$TAG" <<'PY'
import csv, glob, collections, sys
O, tag = sys.argv[1], sys.argv[2]
def short(n): return n.split("(")[0].replace("eip::", "").replace("void ", "").strip()
stats = {}
for f in glob.glob(O + "/ks/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"]))
# a kernel launched with several grids per step (k_msm_accum_l, k_msm_rowcol: the main launch and the top half-window's side
# launch) has an average that belongs to neither: also report the average over the launches with its LARGEST grid
big = {}
for f in glob.glob(O + "/ks/*/*kernel_trace.csv"):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
        by[short(r["Kernel_Name"])][g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, grids in by.items():
        d = grids[max(grids)]
        big[k] = (len(d), sum(d) / len(d))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(O + "/" + sub + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, (calls, avg_ms, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    c = {n: max(v) for n, v in ctr.get(k, {}).items()}          # per launch: the largest dispatch of that kernel
    wc = c.get("SQ_WAVE_CYCLES", 0)
    rows.append([k, calls, "%.4f" % avg_ms, "%.4f" % big.get(k, (0, avg_ms))[1], "%.1f" % pct, "%.0f" % c.get("FETCH_SIZE", -1), "%.0f" % c.get("WRITE_SIZE", -1),
                 "%.0f" % ((max(c.get("FETCH_SIZE", 0), 0) + max(c.get("WRITE_SIZE", 0), 0)) * 1024),
                 "%.0f" % c.get("SQ_WAVES", -1), "%.0f" % c.get("SQ_INSTS_VALU", -1),
                 "%.3f" % (c.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else -1), "%.3f" % (c.get("SQ_WAIT_ANY", 0) / wc if wc else -1),
                 "%.3f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else -1)])
out = "gpurun_out/%s_kernels.csv" % tag
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_ms", "avg_ms_largest_grid", "pct_of_gpu_time", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB", "hbm_bytes_raw_per_launch",
                "SQ_WAVES", "SQ_INSTS_VALU", "valu_active_frac_of_wave_cycles", "wait_any_frac", "wait_inst_frac"])
    w.writerows(rows)
print(open(out).read())
PY
