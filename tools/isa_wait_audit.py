#!/usr/bin/env python3
"""Flags kernels whose global loads look serialised in the ISA: number of `s_waitcnt vmcnt` per global/buffer load.
(A `cond ? p[i] : 0` load can compile to a branch with its own wait)

    python tools/isa_wait_audit.py [file.hip ...]     (default: every csrc/*.hip)
"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal-brain-pattern-identification_xai_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
for path in files:
    extra = ["-ffp-contract=off"] if os.path.basename(path) in ("montage.hip", "specprep.hip") else ["-ffp-contract=fast"]
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", *extra, "-S", "--cuda-device-only", path, "-o", "-"],
                         capture_output=True, text=True).stdout
    parts = re.split(r"\n(_Z\w+|k_\w+):[ \t]*;[^\n]*\n", asm)
    names, bodies = parts[1::2], parts[2::2]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, d, b in zip(names, dem, bodies):
        b = b.split("s_endpgm")[0]
        loads = len(re.findall(r"global_load|buffer_load", b))
        branches = len(re.findall(r"s_cbranch", b))
        # batches: loads issued between two waits that drain the queue completely (vmcnt(0)); batch size 1 = a serial round trip
        batches, cur = [], 0
        for line in b.split("\n"):
            if "global_load" in line or "buffer_load" in line:
                cur += 1
            elif "s_waitcnt" in line and re.search(r"vmcnt\(0\)", line) and cur:
                batches.append(cur)
                cur = 0
        singles = sum(1 for x in batches if x == 1)
        if loads >= 8 and singles >= 4:
            short = re.sub(r"\(.*", "", d).replace("void ", "")
            print("%-14s %-58s loads %4d  full drains %3d  of which after ONE load %3d  branches %4d"
                  % (os.path.basename(path), short[:58], loads, len(batches), singles, branches))
