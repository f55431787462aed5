"""CPU: oracle pieces that have no reference counterpart to pin against (canonical definitions): internal consistency."""
import torch

from oracle import ref_torch as O


def test_expected_gradients_linear_model_is_exact():
    """for a linear model f(x) = <w, x>, phi(x) = (x - E[b]) * w for ANY alpha draws"""
    torch.manual_seed(0)

    class Lin(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(3, 7))

        def forward(self, x):
            return x.flatten(1) @ self.w.t()

    net = Lin()
    x, bg = torch.randn(2, 7), torch.randn(64, 7)
    phi = O.expected_gradients(net, x, bg, nsamples=4000, seed=1)
    want = (x[:, None, :] - bg.mean(0)[None, None, :]) * net.w.detach()[None]
    assert float((phi - want).abs().max() / want.abs().max()) < 0.08     # Monte-Carlo over the baseline choice only


def test_ig_completeness():
    net = O.fill_params(O.build_multimodal(19, 2000, 4, dropout=0.0), seed=51).eval()
    eeg, spec = O.seeded((1, 1, 19, 2000), 52, "randn"), O.seeded((1, 4, 32, 64), 53, "rand")
    ie, is_ = O.integrated_gradients(net, (eeg, spec), n_steps=50)
    with torch.no_grad():
        out = net(eeg, spec); c = int(out.argmax(1))
        base = net(torch.zeros_like(eeg), torch.zeros_like(spec))
    total = float(ie.sum() + is_.sum())
    assert abs(total - float(out[0, c] - base[0, c])) < 0.05 * abs(float(out[0, c] - base[0, c])) + 1e-3
