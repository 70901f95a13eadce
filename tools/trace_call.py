#!/usr/bin/env python3
"""Timeline of the LAST call in a rocprofv3 --kernel-trace --memory-copy-trace run: kernels and copies with their start / end in us
relative to the call's first event, so that what overlaps what is visible.
  tools/trace_call.py <dir> <substring of the call's last kernel> [window_ms]"""
import csv, glob, os, sys
d, last_kernel = sys.argv[1], sys.argv[2]
window = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 8e6
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s " % r.get("Queue_Id", "?") + r["Kernel_Name"].split("(")[0].replace("eip::", "").replace("void ", "")[:44]))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
ev.sort()
ends = [e for e in ev if last_kernel in e[2]]
t_end = ends[-1][1]
sel = [e for e in ev if e[1] <= t_end + 200000 and e[0] >= t_end - window]
t0 = sel[0][0]
for a, b, n in sel:
    print("%9.1f .. %9.1f (%8.1f) %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n))
