#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the per-kernel summaries that tools/prof_kernels.sh writes (rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE in separate passes of the same bench.py command): one entry per (workload, dominant kernel), read by
bench.py for `roofline.traffic`.   usage: tools/make_traffic_json.py r03 g1msm:20:profiles/r03_g1msm_2p20_kernels.csv:k_msm_accum_l ..."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], []
for spec in sys.argv[2:]:
    wl, log2n, path, kernel = spec.split(":")
    for r in csv.DictReader(open(os.path.join(ROOT, path))):
        if r["kernel"].split("<")[0] == kernel:
            fetch, write = float(r["FETCH_SIZE_KiB_raw"]) * 1024, float(r["WRITE_SIZE_KiB"]) * 1024
            out.append({"workload": wl, "log2n": int(log2n), "kernel": kernel, "traffic_bytes_raw": fetch + write,
                        "traffic_bytes_fetch_x2": 2 * fetch + write, "source": os.path.basename(path) +
                        " (tools/prof_kernels.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of the same bench.py "
                        "command, KiB units, largest dispatch of the kernel)",
                        "valu_active_frac_of_wave_cycles": float(r["valu_active_frac_of_wave_cycles"]),
                        "avg_ms_under_rocprof": float(r.get("avg_ms_largest_grid") or r["avg_ms"])})      # the launches with the kernel's largest grid
            break
    else:
        raise SystemExit("kernel %s not found in %s" % (kernel, path))
with open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
