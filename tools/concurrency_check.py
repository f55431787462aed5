#!/usr/bin/env python3
"""Do the two branches give bit-identical results when they run beside each other on two streams?  Forward and backward of
the EEG and spectrogram branches (B=64, bf16, train mode, dropout 0) serially and concurrently; per-tensor max |diff|."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainxai
from brainxai import ops

dev = torch.device("cuda", 0)
B = 64
g = torch.Generator().manual_seed(1)
eeg = torch.randn(B, 1, 19, 2000, generator=g).to(dev)
spec = torch.rand(B, 4, 128, 256, generator=g).to(dev)
torch.manual_seed(9)
m = brainxai.build_multimodal(19, 2000, 4, dropout=0.0, compute_dtype=torch.bfloat16).to(dev).train()
sd = {k: v.clone() for k, v in m.state_dict().items()}
side = torch.cuda.Stream()


def run(concurrent):
    m.load_state_dict(sd)
    m.zero_grad()
    outs = {}
    hooks = [getattr(m.spectrogram_model, f"block{i}").register_forward_hook(lambda mod, i_, o, i=i: outs.__setitem__(f"block{i}", o.detach().float().clone()))
             for i in range(1, 6)]
    cur = torch.cuda.current_stream()
    if concurrent:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ef = m.eeg_model.features(eeg)
            e = ops.LinearLsmFn.apply(ef, m.eeg_model.dense.weight, m.eeg_model.dense.bias)
        s = m.spectrogram_model(spec)
        cur.wait_stream(side)
    else:
        ef = m.eeg_model.features(eeg)
        e = ops.LinearLsmFn.apply(ef, m.eeg_model.dense.weight, m.eeg_model.dense.bias)
        s = m.spectrogram_model(spec)
    for h in hooks:
        h.remove()
    outs["eeg_feat"], outs["eeg_logp"], outs["spec_logp"] = ef.detach().clone(), e.detach().clone(), s.detach().clone()
    r = torch.linspace(-1, 1, 6, device=dev)
    if concurrent:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            (e * r).sum().backward()
        (s * r).sum().backward()
        cur.wait_stream(side)
    else:
        (e * r).sum().backward()
        (s * r).sum().backward()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if p.grad is not None:
            outs["grad." + n] = p.grad.detach().clone()
    return outs


a = run(False)
a2 = run(False)
c = run(True)
bad = 0
for k in a:
    d0 = float((a[k].float() - a2[k].float()).abs().max())
    d1 = float((a[k].float() - c[k].float()).abs().max())
    if d0 != 0 or d1 != 0:
        bad += 1
        print(f"{k:50s} serial-vs-serial {d0:.3e}   serial-vs-concurrent {d1:.3e}   (scale {float(a[k].float().abs().max()):.3e})")
print("tensors compared:", len(a), " differing:", bad)
