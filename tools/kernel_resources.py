#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel of the product object (no GPU needed):
   python tools/kernel_resources.py > profiles/rNN_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "h264-lab_amd", "csrc", "h264e_kernels.hip")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-I" + os.path.join(ROOT, "include"), "-o", os.path.join(d, "k.o")],
                       capture_output=True, text=True)
rows, cur = [], None
pats = (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
        ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"))
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, pat in pats:
        m = re.search(pat, line)
        if m and cur is not None and k not in cur:
            cur[k] = m.group(1)
print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage h264e_kernels.hip  (tools/kernel_resources.py)")
print("# kernel<GEOM, WAVES, OCC>: GEOM 0 wide window / 1 narrow window / 2 intra-only; WAVES per macroblock row; OCC = waves per SIMD aimed at")
print("%-34s %6s %6s %8s %10s %10s %8s %6s" % ("kernel", "VGPRs", "AGPRs", "scratch", "SGPRspill", "VGPRspill", "LDS", "occ"))
for x in rows:
    n = x["name"]
    m = re.match(r"_Z15h264e_mb_kernelILi(\d+)ELi(\d+)ELi(\d+)E", n)
    label = "h264e_mb_kernel<%s,%s,%s>" % m.groups() if m else re.sub(r"^_Z\d+", "", n)[:34]
    print("%-34s %6s %6s %8s %10s %10s %8s %6s" % (label, x.get("vgpr"), x.get("agpr"), x.get("scratch"), x.get("sspill"), x.get("vspill"), x.get("lds"), x.get("occ")))
