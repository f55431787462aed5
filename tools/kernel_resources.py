#!/usr/bin/env python3
"""Print VGPR / AGPR / spill / scratch / LDS / occupancy per kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
`scratch` (bytes per lane) can be non-zero with zero spills: a register array indexed at run time, or filled under a condition,
is placed in scratch memory."""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-c", src,   # (the shipped build's code generation: build.py FLAGS)
                      "-o", "/tmp/_kr.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?)( \[-Rpass)", line) or re.search(r"remark: +(.*?)( \[-Rpass)", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            if pat in cur["name"]:
                print(f"{cur['name'][:60]:60s} VGPR {cur.get('VGPRs','?'):>4s} AGPR {cur.get('AGPRs','?'):>4s} spill {cur.get('VGPR Spill', cur.get('VGPRs Spill','?')):>3s} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} "
                      f"occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} LDS {cur.get('LDS Size [bytes/block]','?')}")
