"""Print the kernel timeline of one call from a rocprofv3 kernel trace: tools/trace_step.py <dir> <first-kernel-substring> [index]"""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) // 2
i0 = starts[n]; i1 = starts[n + 1] if n + 1 < len(starts) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-36s q%-3s start %8.1f end %8.1f dur %8.1f  grid %s" % (r["Kernel_Name"].split("(")[0].replace("eip::", "").replace("void ", "")[:36],
          r.get("Queue_Id", "?"), a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
