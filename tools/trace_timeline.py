#!/usr/bin/env python3
"""Print the kernel timeline (start/end in us relative to the first kernel of the last call) from a
rocprofv3 --kernel-trace csv: shows which kernels overlap."""
import csv
import glob
import sys

pat = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = []
for f in glob.glob(pat):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("eip::", ""), r.get("Queue_Id", "?")))
rows.sort()
rows = rows[-last:]
t0 = rows[0][0]
for a, b, n, q in rows:
    print("%9.1f .. %9.1f us  (%7.1f)  q=%s  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, n[:50]))
