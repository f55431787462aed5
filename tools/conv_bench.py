"""Per-shape timing of the 3x3 conv family (forward / dgrad / wgrad) at the bench workload's stage shapes.

    python tools/conv_bench.py [--iters 50] [--batch 64] [--kinds fwd,dgrad,wgrad]

For each stage shape prints the HIP-event average launch time, the algorithmic TFLOP/s and GB/s, and the two lower
bounds (HBM at 6.3 TB/s achievable, MFMA at 2.5 PFLOP/s dense bf16) so kernel work can be aimed at the worst ratio.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brainxai  # noqa: E402
from brainxai import ops  # noqa: E402

STAGES = [(128, 256, 4, 16), (64, 128, 16, 32), (32, 64, 32, 64), (16, 32, 64, 128), (8, 16, 128, 256)]


def timed(fn, iters):
    """Average GPU time of one call: `iters` calls captured in a hipGraph (no host launch gaps), replayed 3 times."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3       # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="f32: the split (h + m + l, six MFMAs) kernels of fp32 storage")
    a = ap.parse_args()
    kinds = a.kinds.split(",")
    dev, dt, B = torch.device("cuda:0"), (torch.bfloat16 if a.dtype == "bf16" else torch.float32), a.batch
    esz, nmfma = (2, 1) if a.dtype == "bf16" else (4, 6)
    tot = {k: 0.0 for k in kinds}
    print(f"{'shape':26s} {'kind':6s} {'us':>8s} {'TF/s':>8s} {'GB/s':>8s} {'hbm_us':>7s} {'mfma_us':>7s}")
    for (H, W, cin, c) in STAGES:
        for (ci, co) in ((cin, c), (c, c)):
            cip, cop = ops.pad8(ci), ops.pad8(co)
            x = torch.randn(B, H, W, cip, device=dev).to(dt)
            dz = torch.randn(B, H, W, cop, device=dev).to(dt)
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
            bias = torch.zeros(co, device=dev)
            pk_f = ops._pack(w, flip=False, dtype=dt)
            pk_b = ops._pack(w, flip=True, dtype=dt)
            y = torch.randn(B, H, W, cip, device=dev).to(dt)
            flops = 2.0 * B * H * W * 9 * cip * cop
            byts = float(esz) * B * H * W * (cip + cop)
            mult = 2 if (ci, co) == (c, c) else 1     # the stage has two c->c convs
            for k in kinds:
                if k == "fwd":
                    t = timed(lambda: ops._conv(x, pk_f, bias, None, None, True, dt), a.iters)
                elif k == "dgrad":
                    if ci == cin and cin == 4:
                        continue                        # the input layer needs no data gradient
                    t = timed(lambda: ops._conv(dz, pk_b, None, y, None, False, dt), a.iters)
                else:
                    ops.clear_grad_views()
                    t = timed(lambda: ops._wgrad(x, dz, w, bias), a.iters)
                tot[k] += t * mult
                print(f"B{B} {H}x{W} {cip:3d}->{cop:3d} x{mult}   {k:6s} {t:8.1f} {flops / t * 1e-6:8.1f} {byts / t * 1e-3:8.1f} "
                      f"{byts / 6.3e6:7.1f} {nmfma * flops / 2.5e9:7.1f}")
    print("per-step totals (us):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))


if __name__ == "__main__":
    main()
