#!/usr/bin/env python3
"""Diagnostic for BX_OVERLAP_EEG: is the FORWARD of the overlapped step (EEG branch on a side stream) reproducible when replayed from a
hipGraph?  Captures the two branches' training-mode forward alone and compares every replay's branch outputs with the serial eager ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainxai
from brainxai import ops

dev = torch.device("cuda", 0)
B = 64
g = torch.Generator().manual_seed(1)
eeg, spec = torch.randn(B, 1, 19, 2000, generator=g).to(dev), torch.rand(B, 4, 128, 256, generator=g).to(dev)
torch.manual_seed(9)
m = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
for p in m.parameters():
    p.requires_grad_(False)
for p in m.eeg_model.parameters():
    p.requires_grad_(True)            # keeps the EEG branch's saved arena reachable (grad_fn.saved_tensors)
sm, em = m.spectrogram_model, m.eeg_model
MODE = os.environ.get("MODE", "both")          # both | eeg_only | spec_only (which branch runs inside the fork)


def fwd(overlap):
    with torch.enable_grad():
        xi, ss, se = sm._pack_all(spec, seed_pair=True)
        if not overlap:
            return em.features(eeg, seed=se), sm.features(spec, seed=ss, packed=(xi,))
        cur = torch.cuda.current_stream()
        side = ops.side_stream("eeg", dev)
        side.wait_stream(cur)
        se.record_stream(side)
        with torch.cuda.stream(side):
            ef = em.features(eeg, seed=se) if MODE != "spec_only" else torch.zeros(1, device=dev)
        if MODE.startswith("blocks"):                       # only some of the spectrogram stages beside the EEG branch, e.g. blocks1-2, blocks3-5
            lo, hi = (int(v) for v in MODE[6:].split("-"))
            xs = xi if lo == 1 else SYN[lo]
            blocks = [getattr(sm, f"block{i}") for i in range(1, 6)]
            for b_ in blocks:
                b_._seed = ss
            sf = xs
            for i in range(lo, hi + 1):
                sf = blocks[i - 1](sf)
        else:
            sf = sm.features(spec, seed=ss, packed=(xi,)) if MODE != "eeg_only" else torch.zeros(1, device=dev)
        cur.wait_stream(side)
        ef.record_stream(cur)
        return ef, sf


# synthetic inputs for starting the spectrogram branch at a later stage (internal layout: logical NCHW view of channels-last bf16)
SYN = {i: (torch.rand(B, 128 >> (i - 1), 256 >> (i - 1), c, device=dev) - 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
       for i, c in ((2, 16), (3, 32), (4, 64), (5, 128))}
ops.manual_seed(1234)
_m, MODE = MODE, "both"
r0 = fwd(False)
MODE = _m
ref = [t.detach().clone() for t in r0]
saved_ref = r0[0].grad_fn.saved_tensors[1].clone()
print("saved arena bytes", saved_ref.numel())
for _ in range(3):
    fwd(True)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = fwd(True)
saved_g = out[0].grad_fn.saved_tensors[1]
out = [t.detach() for t in out]
bad = [0, 0]
N = int(os.environ.get("N", "200"))
for i in range(N):
    ops.manual_seed(1234)
    graph.replay()
    torch.cuda.synchronize()
    if not torch.equal(saved_g, saved_ref) and bad[0] < 3:
        db = (saved_g != saved_ref).nonzero().flatten()
        w = db // 4
        print("  saved arena: differing bytes", db.numel(), "first", int(db[0]), "last", int(db[-1]))
        # coarse map: 64 KB bins with differing bytes
        lo = db[db < saved_ref.numel() - 4096]
        for w4 in torch.unique(lo // 4).tolist()[:6]:
            a = saved_g[4 * w4:4 * w4 + 4].clone().view(torch.float32).item(); b_ = saved_ref[4 * w4:4 * w4 + 4].clone().view(torch.float32).item()
            print("    word", w4, "bytes differing", [int(x) % 4 for x in lo[(lo // 4) == w4].tolist()], "graph", a, "serial", b_,
                  "neighbours graph", saved_g[4 * w4 - 8:4 * w4 + 12].clone().view(torch.float32).tolist())
        bins = torch.unique(db // 65536)
        print("  64KB bins touched:", bins.tolist()[:40], "..." if bins.numel() > 40 else "")
    for k in range(2):
        if out[k].shape == ref[k].shape and not torch.equal(out[k], ref[k]):
            bad[k] += 1
            if bad[k] <= 2:
                d = (out[k].float() - ref[k].float()).abs()
                print("replay", i, "EEG features" if k == 0 else "spectrogram features", "max diff", float(d.max()), "elements differing", int((d > 0).sum()), "of", d.numel())
                if k == 0:
                    dd = (d > 0).reshape(d.shape[0], 16, -1)          # [B, F2, T2]
                    print("   per sample:", dd.sum((1, 2)).tolist())
                    print("   per channel:", dd.sum((0, 2)).tolist())
                    print("   per time step:", dd.sum((0, 1)).tolist())
print(f"MODE={MODE}: replays with differing EEG features {bad[0]}, spectrogram features {bad[1]} of {N}")
