#!/usr/bin/env python3
"""In-kernel timeline of k_wgrad_own (diagnostic build, never the shipped library): compiles conv3x3_mfma.hip with
-DBX_WGRAD_STAMPS into a scratch .so, runs one weight-gradient launch per shape and prints, for the sampled workgroups, the
shader-clock deltas between: start -> first tile staged -> end of tile 0..3 -> partial stored (s_memtime ticks = shader cycles).

    python tools/wgrad_stamps.py            (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "multimodal-brain-pattern-identification_xai_amd")
out = "/tmp/libbrainxai_stamps.so"
import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("_bx_build", os.path.join(PKG, "build.py")); _b = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_b)
srcs = _b.SOURCES
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DBX_WGRAD_STAMPS", "-shared",
                       *[os.path.join(PKG, "csrc", f) for f in srcs], "-o", out])
import torch  # noqa: E402
import brainxai  # noqa: E402
from brainxai import _lib as L, ops  # noqa: E402
L.LIB_PATH = out
L._lib = None
lib = L.load()
lib.bx_debug_wgrad_stamps.restype = C.c_int
lib.bx_debug_wgrad_stamps.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
for (H, W, ci, co) in ((64, 128, 32, 32), (32, 64, 64, 64), (16, 32, 128, 128), (8, 16, 256, 256)):
    B = 64
    x = torch.randn(B, H, W, ci, device=dev).bfloat16(); dz = torch.randn(B, H, W, co, device=dev).bfloat16()
    w = torch.randn(co, ci, 3, 3, device=dev); bias = torch.zeros(co, device=dev)
    for _ in range(3):
        ops._wgrad(x, dz, w, bias)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 512)()
    assert lib.bx_debug_wgrad_stamps(C.cast(buf, C.c_void_p)) == 0
    rows = [[buf[i * 8 + k] for k in range(7)] for i in range(64)]
    rows = [r for r in rows if r[0] and r[6] > r[0]]
    t0 = min(r[0] for r in rows)
    print(f"== {H}x{W} {ci}->{co}: {len(rows)} sampled workgroups; columns: start offset | stage0 | tile0 | tile1 | tile2 | tile3 | store (cycles)")
    for r in rows[:12]:
        d = [r[0] - t0] + [r[k + 1] - r[k] if r[k + 1] and r[k] else 0 for k in range(6)]
        print("   " + " ".join(f"{v:8d}" for v in d))
    life = sorted(r[6] - r[0] for r in rows)
    print(f"   workgroup lifetime: median {life[len(life) // 2]} cycles, max {life[-1]}; span of sampled starts {max(r[0] for r in rows) - t0}, last end - first start {max(r[6] for r in rows) - t0}")
