#!/bin/bash
# Compact per-kernel resource table (VGPR / spills / scratch / LDS / occupancy) for the HIP library.
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Imoka_amd/csrc -Wno-unused-function moka_amd/csrc/moka_kernels.hip -o /tmp/_moka_res.so \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip().split("(")[0]};rows.append(cur);continue
    for k,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("spill",r"VGPRs Spill: (\d+)"),("sgpr",r" SGPRs: (\d+)")):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
print("%-62s %5s %5s %6s %7s %4s"%("kernel","vgpr","sgpr","spill","scratch","occ"))
for r in rows: print("%-62s %5s %5s %6s %7s %4s"%(r["name"][-62:],r.get("vgpr"),r.get("sgpr"),r.get("spill"),r.get("scratch"),r.get("occ")))
'
