#!/usr/bin/env python3
"""Where does the bf16 Grad-CAM sweep's distance from the fp32 oracle come from?  Same weights, same 16 samples: block5 activation A,
channel weights w = mean_hw dy_c/dA, raw map sum_c w_c A_c, each against the CPU fp32 oracle, for bf16 and fp32 storage, at random
initialisation and after `--train` optimizer steps on the batch (what bench.py's model looks like when its sweep runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import brainxai  # noqa: E402
from oracle import ref_torch as O  # noqa: E402

train = int(sys.argv[sys.argv.index("--train") + 1]) if "--train" in sys.argv else 0
dev = torch.device("cuda:0")
B = 16
g = torch.Generator().manual_seed(42)
spec = torch.rand(B, 4, 128, 256, generator=g)
eeg = torch.randn(B, 1, 19, 2000, generator=g)
labels = torch.softmax(torch.randn(B, 6, generator=g), 1)
torch.manual_seed(42)
ref = O.build_multimodal(19, 2000, 4, dropout=0.5)
if train:
    net = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
    net.load_state_dict(ref.state_dict())
    opt = brainxai.FlatAdamW(net.parameters(), lr=1e-3)
    for _ in range(train):
        brainxai.train_step(net, opt, eeg.to(dev), spec.to(dev), labels.to(dev), brainxai.KLDivLoss())
    torch.cuda.synchronize()
    ref.load_state_dict(net.state_dict())
    opt.close()
ref.eval()
cam_o, raw_o, w_o, A_o, out_o = O.grad_cam(ref, eeg, spec, "spectrogram_model.block5", "all", upsample=False, return_parts=True)
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
print(f"train steps {train}; oracle: |A| max {float(A_o.abs().max()):.3g}  |w| max {float(w_o.abs().max()):.3g}  |raw| max {float(raw_o.abs().max()):.3g}  "
      f"sum_c |w_c A_c| max {float((w_o.abs()[:, :, :, None, None] * A_o.abs()[:, None]).sum(2).max()):.3g}")
for dt in (torch.float32, torch.bfloat16):
    net = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=dt).to(dev)
    net.load_state_dict(ref.state_dict())
    net.eval()
    cam, raw, w, A, out = brainxai.grad_cam(net, eeg.to(dev), spec.to(dev), "spectrogram_model.block5", "all", upsample=False, return_parts=True)
    if A.shape != A_o.shape:
        A = A.permute(0, 3, 1, 2)                      # the product hands the stage output out channels-last
    sw = brainxai.GradCamSweep(net, eeg.to(dev), spec.to(dev), class_idx="all", upsample=False)
    cs = sw(eeg.to(dev), spec.to(dev))
    print(f"{str(dt):15s} logits {rel(out, out_o):.2e}  A {rel(A.float(), A_o):.2e}  w {rel(w, w_o):.2e}  raw {rel(raw, raw_o):.2e}  "
          f"cam(sweep) vs oracle on the raw scale {float((cs.double().cpu() - cam_o.double()).abs().max() / raw_o.double().abs().max()):.2e}")
