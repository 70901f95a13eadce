#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS of the built library: tools/kernel_resources.py [libeip2537_hip.so]
(llvm-objdump --offloading, then the AMDGPU metadata notes of every gfx950 code object)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
so = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "blst_eip2537_amd", "libeip2537_hip.so"))
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, os.path.basename(so))
    os.symlink(so, tmp)
    subprocess.check_call([LLVM + "llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, cwd=d)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", os.path.join(d, f)], stdout=subprocess.PIPE, text=True).stdout
        for b in notes.split("- .agpr_count")[1:]:
            g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
            name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
            print("%-58s vgpr %3s sgpr %3s scratch %5s spill %3s lds %6s" % (name.replace("eip::", "")[:58], g("vgpr_count"), g("sgpr_count"),
                  g("private_segment_fixed_size"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
