"""Debug: per-step parity with the oracle re-synchronised to the product's weights before every step (separates
kernel errors from the chaotic drift of AdamW trajectories)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import brainxai
from brainxai import ops
from oracle import ref_torch as O
DEV = torch.device("cuda:0")
ref = O.fill_params(O.build_multimodal(19, 2000, 4, dropout=0.0), seed=71)
mine = brainxai.build_multimodal(19, 2000, 4, dropout=0.0); mine.load_state_dict(ref.state_dict()); mine.to(DEV)
opt = brainxai.FlatAdamW(mine.parameters(), lr=1e-3)
opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-3)
b = 8
for step in range(4):
    seed = 500 + 10 * step
    eeg = O.seeded((b, 1, 19, 2000), seed, "randn"); spec = O.seeded((b, 4, 64, 128), seed + 1, "rand")
    lab = F.one_hot(torch.randint(0, 6, (b,), generator=torch.Generator().manual_seed(seed + 2)), 6).float()
    ref.load_state_dict({k: v.cpu() for k, v in mine.state_dict().items()})
    ref.train(); mine.train()
    opt_r.zero_grad(); opt.zero_grad()
    out = ref(eeg, spec); loss = torch.nn.KLDivLoss()(out, lab); loss.backward()
    o2 = mine(eeg.to(DEV), spec.to(DEV)); l2 = brainxai.KLDivLoss()(o2, lab.to(DEV)); l2.backward()
    gscale = max(float(p.grad.abs().max()) for p in ref.parameters())
    worst = ("", 0.0)
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        err = float((p.grad - q.grad.cpu()).abs().max()) / max(float(p.grad.abs().max()), 1e-3 * gscale)
        if err > worst[1]: worst = (n, err)
    print(f"step {step}: loss ref {float(loss):.6f} mine {float(l2):.6f}  out maxdiff {float((out - o2.cpu()).abs().max()):.2e}  worst grad {worst}")
    opt.step()
    ref.eval(); mine.eval()
    with torch.no_grad():
        ref.load_state_dict({k: v.cpu() for k, v in mine.state_dict().items()})
        oe = ref(eeg, spec); om = mine(eeg.to(DEV), spec.to(DEV))
        print(f"         eval out maxdiff {float((oe - om.cpu()).abs().max()):.2e}  |out| {float(oe.abs().max()):.2f}")
ops.clear_grad_views()
