import sys, torch
sys.path.insert(0, "/root/repo")
import brainxai
from brainxai import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, T = 64, 128, 256, 2000
model = brainxai.build_multimodal(19, T, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
step = brainxai.GraphedTrainStep(model, opt, brainxai.KLDivLoss())
g = torch.Generator(device=dev).manual_seed(1)
eeg = torch.randn(B, 1, 19, T, generator=g, device=dev); spec = torch.rand(B, 4, H, W, generator=g, device=dev)
y = torch.softmax(torch.randn(B, 6, generator=g, device=dev), 1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    loss, _ = step((eeg, spec), y)
print("loss", float(loss))
model.eval()
se = torch.randn(256, 1, 19, T, generator=g, device=dev); ss = torch.rand(256, 4, H, W, generator=g, device=dev)
sweep = brainxai.GradCamSweep(model, se[:B], ss[:B], class_idx="all")
for b0 in range(0, 256, B):
    m = sweep(se[b0:b0 + B], ss[b0:b0 + B]).clone()
    w = brainxai.grad_cam(model, se[b0:b0 + B], ss[b0:b0 + B], class_idx="all")
    raw = brainxai.grad_cam(model, se[b0:b0 + B], ss[b0:b0 + B], class_idx="all", relu=False)
    print(b0, "sweep sum", float(m.sum()), "eager sum", float(w.sum()), "equal", torch.equal(m, w), "raw min/max", float(raw.min()), float(raw.max()))
