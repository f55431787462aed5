#!/usr/bin/env python3
"""Where does the fp32 saliency map leave the decision-matched fp64 oracle?  (debugging aid for tests/test_gpu_parity.py::test_saliency_and_ig)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import brainxai  # noqa: E402
from brainxai import ops  # noqa: E402
from oracle import ref_torch as O  # noqa: E402
from tests.golden_util import matched_oracle  # noqa: E402
DEV = torch.device("cuda:0")
ref = O.fill_params(O.build_multimodal(19, 2000, 4, dropout=0.0), seed=51).eval()
mine = brainxai.build_multimodal(19, 2000, 4, dropout=0.0); mine.load_state_dict(ref.state_dict()); mine.to(DEV).eval()
eeg, spec = O.seeded((2, 1, 19, 2000), 52, "randn"), O.seeded((2, 4, 64, 128), 53, "rand")
keep = ops.keep_block_activations(mine)
grads_m, grads_t = {}, {}
def hook_m(name):
    def f(mod, inp, out):
        out.register_hook(lambda g, name=name: grads_m.__setitem__(name, g.detach().float().cpu().double()))
    return f
def hook_t(name):
    def f(mod, inp, out):
        out.register_hook(lambda g, name=name: grads_t.__setitem__(name, g.detach().double()))
    return f
for i in range(1, 6):
    getattr(mine.spectrogram_model, f"block{i}").register_forward_hook(hook_m(f"block{i}"))
e = eeg.to(DEV).requires_grad_(True); s = spec.to(DEV).requires_grad_(True)
for p in mine.parameters(): p.requires_grad_(False)
out = mine(e, s)
seed = torch.zeros_like(out); seed[torch.arange(2), out.argmax(1)] = 1
ge, gs = torch.autograd.grad(out, (e, s), grad_outputs=seed)
torch.cuda.synchronize()
twin, flips = matched_oracle(O, ref, (eeg, spec), keep, "diag")
for i in range(1, 6):
    getattr(twin.spectrogram_model, f"block{i}").register_forward_hook(hook_t(f"block{i}"))
ed, sd = eeg.double().requires_grad_(True), spec.double().requires_grad_(True)
ot = twin(ed, sd)
print("out err", float((out.detach().cpu().double() - ot.detach()).abs().max()), "decision log", twin.decision_log)
gte, gts = torch.autograd.grad(ot, (ed, sd), grad_outputs=seed.cpu().double())
d = (gs.cpu().double() - gts).abs()
print("spec grad: max abs err", float(d.max()), "scale", float(gts.abs().max()), "rel", float(d.max() / gts.abs().max()))
idx = (d > 2e-4 * gts.abs().max()).nonzero()
print("outliers", idx.shape[0], idx[:20].tolist())
for i in idx[:10].tolist():
    print(i, float(gs.cpu()[tuple(i)]), float(gts[tuple(i)]))
de = (ge.cpu().double() - gte).abs()
print("eeg grad rel", float(de.max() / gte.abs().max()))

for k in sorted(grads_t):
    a, b = grads_m[k], grads_t[k]
    print(k, "grad at stage output: rel max err", float((a - b).abs().max() / b.abs().max()), "rel L2", float((a - b).norm() / b.norm()), "ratio of norms", float(a.norm() / b.norm()))
