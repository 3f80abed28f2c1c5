#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed)."""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "altair-raytracing_amd", "csrc")
out = subprocess.run(["make", "-s", "-C", here, "resource-usage"] + sys.argv[1:], capture_output=True, text=True).stderr
cur, d = None, {}
for l in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(": ")[1]; d[cur] = {}
    elif ": " in t and cur:
        k, v = t.split(": ", 1); d[cur][k.strip()] = v
for k, v in d.items():
    print(f"{k:34s} VGPR {v.get('VGPRs'):>4s} AGPR {v.get('AGPRs'):>3s} scratch {v.get('ScratchSize [bytes/lane]'):>4s} "
          f"occ {v.get('Occupancy [waves/SIMD]')} SGPR {v.get('TotalSGPRs')} spillS {v.get('SGPRs Spill','-')} spillV {v.get('VGPRs Spill','-')} LDS {v.get('LDS Size [bytes/block]')}")
