import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, brainxai
from oracle import ref_torch as O
DEV = torch.device("cuda:0")
def run(cin, c, h, w, kind, mode):
    ref = O.fill_params(O.Block(cin, c, kind, (2, 2), dropout_p=0.0), seed=7)
    mine = brainxai.Block(cin, c, kind, (2, 2), dropout_p=0.0); mine.load_state_dict(ref.state_dict()); mine.to(DEV)
    x = O.seeded((2, cin, h, w), 11, "randn"); r = O.seeded((2, c, h // 2, w // 2), 12, "randn")
    ref.train(mode == "train"); mine.train(mode == "train")
    xr = x.clone().requires_grad_(True); (ref(xr) * r).sum().backward()
    xm = x.clone().to(DEV).requires_grad_(True); (mine(xm) * r.to(DEV)).sum().backward()
    d = (xm.grad.cpu() - xr.grad).abs()
    print(f"case {cin},{c},{h}x{w},{kind},{mode}: max err {float(d.max()):.3e} of {float(xr.grad.abs().max()):.3e}")
    idx = (d > 1e-4 * float(xr.grad.abs().max())).nonzero()
    print("  bad count", len(idx), "of", d.numel())
    if len(idx):
        print("  rows:", sorted(set(idx[:, 2].tolist()))[:40])
        print("  cols:", sorted(set(idx[:, 3].tolist()))[:40])
        print("  chans:", sorted(set(idx[:, 1].tolist())))
    for (n, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        e = float((p.grad.cpu() - q.grad).abs().max() / (q.grad.abs().max() + 1e-30))
        if e > 1e-4: print("  param", n, f"{e:.3e}")
for case in [(3, 16, 50, 37, "max", "eval"), (4, 16, 50, 37, "max", "eval"), (4, 16, 50, 36, "max", "eval"), (4, 16, 51, 36, "max", "eval"),
             (64, 128, 25, 18, "avg", "eval"), (64, 128, 25, 18, "avg", "train")]:
    run(*case)
