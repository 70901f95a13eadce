#!/usr/bin/env python3
"""profiles/rNN_traffic.json from a tools/prof_kernels.sh summary: the PMC traffic of the dominant kernel
(largest share of GPU time) of one workload, in the form bench.py quotes as roofline.traffic.
usage: tools/traffic_json.py <kernels.csv> <workload> <log2n> <out.json>"""
import csv
import json
import os
import sys

src, wl, log2n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = list(csv.DictReader(open(src)))
top = max(rows, key=lambda r: float(r["pct_of_gpu_time"]))
fetch, write = float(top["FETCH_SIZE_KiB_raw"]) * 1024, float(top["WRITE_SIZE_KiB"]) * 1024
json.dump({
    "workload": wl, "log2n": log2n, "kernel": top["kernel"],
    "traffic_bytes_raw": fetch + write, "traffic_bytes_fetch_x2": 2 * fetch + write,
    "source": "%s (tools/prof_kernels.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of the same "
              "bench.py command, KiB units, largest dispatch of the kernel)" % os.path.basename(src),
    "valu_active_frac_of_wave_cycles": float(top["valu_active_frac_of_wave_cycles"]),
    "avg_ms_under_rocprof": float(top["avg_ms"]),
}, open(out, "w"), indent=1)
print(open(out).read())
