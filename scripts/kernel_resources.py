#!/usr/bin/env python3
"""Register / spill / scratch summary of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py uemda_amd/csrc/conv.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off",
       f"-I{ROOT}/include", f"-I{ROOT}/uemda_amd/csrc", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src,
       "-o", "/dev/null"]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in t.split("\n"):
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    print(f"{dem[:100]:100s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} SGPR {r.get('TotalSGPRs'):>3s} vspill {r.get('VGPRs Spill'):>3s} "
          f"sspill {r.get('SGPRs Spill'):>3s} scratch {r.get('ScratchSize'):>4s} occ {r.get('Occupancy')}")
