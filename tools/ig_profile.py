#!/usr/bin/env python3
"""configs[4] under the profiler: integrated gradients, 50 steps x B=64 (bf16 storage), one call after a warm-up call.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 $REPO/tools/ig_profile.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import brainxai
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    B = int(os.environ.get("IG_B", "64"))
    g = torch.Generator().manual_seed(42)
    spec = torch.rand(B, 4, 128, 256, generator=g).to(dev)
    eeg = torch.randn(B, 1, 19, 2000, generator=g).to(dev)
    model = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).eval()
    mb = int(os.environ.get("IG_MAX_BATCH", "1024"))
    brainxai.integrated_gradients(model, (eeg, spec), None, n_steps=4, max_batch=mb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ie, is_ = brainxai.integrated_gradients(model, (eeg, spec), None, n_steps=50, max_batch=mb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"IG 50 steps x B={B}: {dt * 1e3:.1f} ms  -> {B / dt:.1f} samples/s  (checksum {float(is_.abs().sum()):.6g})")


if __name__ == "__main__":
    main()
