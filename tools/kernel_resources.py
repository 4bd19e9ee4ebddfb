#!/usr/bin/env python3
"""Print per-kernel VGPR / SGPR / LDS / scratch / occupancy for the gfx950 build (hipcc remarks)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "oclradixsort_amd", "csrc", "adlhip.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/libadlhip_res.so", src] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("adlhip::", "").replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+[A-Za-z])(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-70s %5s %5s %6s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for k, v in rows.items():
    print("%-70s %5s %5s %6s %7s %4s" % (k[:70], v.get("VGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize"), v.get("LDS Size"), v.get("Occupancy")))
