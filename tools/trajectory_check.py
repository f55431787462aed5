#!/usr/bin/env python3
"""Bench-size (B=64, bf16) trajectory check: eager steps vs hipGraph replay must give the same losses and parameters
(every kernel is deterministic; a difference means a race or a stale buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainxai
from brainxai import ops

ops.OVERLAP_EEG_ENV = ops.OVERLAP_EEG
dev = torch.device("cuda", 0)
B, STEPS = int(os.environ.get("TC_B", "64")), int(os.environ.get("TC_STEPS", "8"))
g = torch.Generator().manual_seed(1)
batches = [((torch.randn(B, 1, 19, 2000, generator=g).to(dev), torch.rand(B, 4, 128, 256, generator=g).to(dev)),
            torch.softmax(torch.randn(B, 6, generator=g), 1).to(dev)) for _ in range(STEPS)]
crit = brainxai.KLDivLoss()
res = {}
modes = ("eager", "graph") + (("serial",) if ops.OVERLAP_EEG else ())     # serial: the same kernels without the EEG side stream
for mode in modes:
    ops.OVERLAP_EEG = ops.OVERLAP_EEG_ENV if mode != "serial" else 0
    torch.manual_seed(9)
    m = brainxai.build_multimodal(19, 2000, 4, dropout=float(os.environ.get("TC_DROPOUT", "0.5")), compute_dtype=torch.bfloat16).to(dev).train()
    opt = brainxai.FlatAdamW(m.parameters(), lr=1e-3)
    ops.manual_seed(1234)
    step = brainxai.GraphedTrainStep(m, opt, crit)
    step.enabled = mode == "graph"
    losses = [float(step(list(x), y)[0]) for x, y in batches]
    torch.cuda.synchronize()
    res[mode] = (losses, torch.cat([p.detach().flatten() for p in m.parameters()]).clone())
    ops.clear_grad_views()
    print(mode, " ".join(f"{v:.6f}" for v in losses[-8:]))
ok = True
ref = "serial" if "serial" in res else "eager"
for mode in [m for m in modes if m != ref]:
    dl = max(abs(a - b) for a, b in zip(res[mode][0], res[ref][0]))
    dp = float((res[mode][1] - res[ref][1]).abs().max())
    print(f"{mode}: max |loss - {ref}| = {dl:.3e}, max |param - {ref}| = {dp:.3e}")
    ok &= dl < 1e-6 and dp < 1e-6
sys.exit(0 if ok else 1)
