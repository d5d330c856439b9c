#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/SGPR/spills/scratch/LDS/occupancy) for one .hip file.
usage: python tools/kres.py dis-pu_amd/csrc/knn.hip [filter-substring]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": v}
    else:
        cur[k.split(" [")[0]] = v
if cur: rows.append(cur)
print("%-70s %5s %5s %5s %6s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "scratch", "LDS", "occ"))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE).stdout.decode().strip()
    name = re.sub(r"\(.*", "", name).replace("void dispu::", "")
    if flt and flt not in name: continue
    print("%-70s %5s %5s %5s %6s %7s %6s %4s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
