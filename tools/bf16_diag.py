#!/usr/bin/env python3
"""Where does the bf16-storage training step leave the fp32 oracle?  Per-tensor gradient cosine / relative L2 of
(a) GPU fp32 storage vs oracle, (b) GPU bf16 storage vs oracle, (c) GPU bf16 vs GPU fp32, at the benchmark configuration on
bf16-exact inputs and conv weights; plus the MFMA weight-gradient kernel alone at the block1 benchmark shape against a
PyTorch fp32 reference on the same bf16-exact operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import brainxai  # noqa: E402
from brainxai import _lib as L, ops  # noqa: E402
from oracle import ref_torch as O  # noqa: E402

DEV = torch.device("cuda:0")
B, CIN, H, W, CHANS, T = 64, 4, 128, 256, 19, 2000
bs = int(sys.argv[1]) if len(sys.argv) > 1 else B


def bf(t):
    return t.bfloat16().float()


def per_op():
    lib = L.load()
    for (cin, cout, h, w, b) in ((16, 16, 128, 256, 64), (16, 16, 128, 256, 8), (32, 32, 64, 128, 64), (256, 256, 8, 16, 64)):
        torch.manual_seed(1)
        x = bf(torch.randn(b, cin, h, w)); dz = bf(torch.randn(b, cout, h, w) * 1e-3)
        wt = torch.zeros(cout, cin, 3, 3, requires_grad=True); bias = torch.zeros(cout, requires_grad=True)
        (F.conv2d(x.double(), wt.double(), bias.double(), padding=1) * dz.double()).sum().backward()
        xn, dzn = ops.to_nhwc(x.to(DEV), torch.bfloat16), ops.to_nhwc(dz.to(DEV), torch.bfloat16)
        for name, algo in (("mfma", L.BX_ALGO_MFMA),):
            need = lib.bx_conv3x3_wgrad_workspace(b, h, w, cin, cout, L.BX_BF16, algo)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
            dw = torch.empty(cout, cin, 3, 3, device=DEV); db = torch.empty(cout, device=DEV)
            L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), b, h, w, cin, cin, cout, L.BX_BF16, algo,
                                         ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "wgrad")
            torch.cuda.synchronize()
            e = float((dw.cpu().double() - wt.grad.double()).abs().max() / wt.grad.abs().max())
            eb = float((db.cpu().double() - bias.grad.double()).abs().max() / bias.grad.abs().max())
            print(f"per-op wgrad {name} cin={cin} cout={cout} {h}x{w} B={b}: dW rel err {e:.2e}  db rel err {eb:.2e}", flush=True)


def model_level():
    syn = O.synthetic_batch(batch=bs, in_channels=CIN, height=H, width=W, chans=CHANS, seed=42, stacked=False)
    eeg = bf(brainxai.stack_eeg(syn["raw_eeg"].to(DEV)).cpu()); spec = bf(syn["spec"]); labels = syn["labels"]
    ref = O.fill_params(O.build_multimodal(CHANS, T, CIN, dropout=0.0), seed=5)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 4:
                p.copy_(bf(p))
    ref.train()
    O.kl_div(ref(eeg, spec), labels).backward()
    g = {}
    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.0, compute_dtype=dt)
        m.load_state_dict(ref.state_dict()); m.to(DEV).train()
        out = m(eeg.to(DEV), spec.to(DEV)); brainxai.KLDivLoss()(out, labels.to(DEV)).backward()
        torch.cuda.synchronize()
        g[tag] = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
    print(f"{'tensor':48s} {'fp32/oracle cos':>16s} {'bf16/oracle cos':>16s} {'bf16/oracle L2':>15s} {'bf16/fp32 L2':>13s}")
    for n, q in ref.named_parameters():
        if q.numel() < 256:
            continue
        o = q.grad.double().flatten()
        a, b_ = g["fp32"][n].flatten(), g["bf16"][n].flatten()
        cos = lambda u, v: float(F.cosine_similarity(u, v, dim=0))      # noqa: E731
        print(f"{n:48s} {cos(a, o):16.6f} {cos(b_, o):16.6f} {float((b_ - o).norm() / o.norm()):15.3e} {float((b_ - a).norm() / a.norm()):13.3e}", flush=True)


per_op()
model_level()
