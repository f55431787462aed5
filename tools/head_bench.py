#!/usr/bin/env python3
"""Times the fused multimodal head alone (hot caches): 50 forward launches / 50 fwd+bwd pairs captured in a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainxai
from brainxai import ops

dev = torch.device("cuda", 0)
B = 64
feat = torch.randn(B, 4, 8, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
ef = torch.randn(B, 992, device=dev, requires_grad=True)
P = lambda *s: torch.randn(*s, device=dev, requires_grad=True) * 0.05
fcw, fcb, dw, db, w1, b1, w2, b2 = P(6, 256), P(6), P(6, 992), P(6), P(128, 12), P(128), P(6, 128), P(6)
args = [t.detach().requires_grad_(True) for t in (fcw, fcb, dw, db, w1, b1, w2, b2)]


def fwd():
    return ops.MultimodalHeadFn.apply(feat, ef, *args)


def fwd_bwd():
    y = fwd()
    torch.autograd.grad(y.sum(), [feat, ef] + args)


for name, fn in (("fwd", fwd), ("fwd+bwd", fwd_bwd)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per iteration")
