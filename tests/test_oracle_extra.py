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


def test_region_stacker_resize_restatement():
    """Row H, spectrogram half: the oracle's skimage.transform.resize restatement (two scipy.ndimage calls, as scikit-image 0.24
    does for 2-D input) against the closed form the HIP kernel implements -- separable gaussian with mirrored indices, then a
    bilinear sample at ((o + 0.5) n_in / n_out - 0.5) with mirrored indices -- up- and down-scaling."""
    import numpy as np
    from oracle import ref_torch as O

    def mir(i, n):
        i = -i if i < 0 else i
        return 2 * (n - 1) - i if i > n - 1 else i

    def manual(x, H, W):
        h, w = x.shape
        for axis, wts in ((0, O.resize_gaussian_weights(h, H)), (1, O.resize_gaussian_weights(w, W))):
            r = len(wts) // 2
            if r:
                xp = np.pad(x, [(r, r) if a == axis else (0, 0) for a in range(2)], mode="reflect")
                x = sum(wts[k] * (xp[k:k + h] if axis == 0 else xp[:, k:k + w]) for k in range(len(wts)))
        out = np.zeros((H, W))
        for oy in range(H):
            cy = (oy + 0.5) * h / H - 0.5; y0 = int(np.floor(cy)); ty = cy - y0
            for ox in range(W):
                cx = (ox + 0.5) * w / W - 0.5; x0 = int(np.floor(cx)); tx = cx - x0
                out[oy, ox] = ((1 - ty) * ((1 - tx) * x[mir(y0, h), mir(x0, w)] + tx * x[mir(y0, h), mir(x0 + 1, w)])
                               + ty * ((1 - tx) * x[mir(y0 + 1, h), mir(x0, w)] + tx * x[mir(y0 + 1, h), mir(x0 + 1, w)]))
        return out
    g = np.random.default_rng(0)
    for (h, w, H, W) in ((100, 300, 128, 256), (100, 300, 32, 100), (37, 53, 64, 20)):
        x = g.random((h, w))
        assert np.abs(O.skimage_resize(x, (H, W)) - manual(x, H, W)).max() < 1e-12, (h, w, H, W)
    fr = O.synthetic_spectrogram_frames(batch=1, trows=320, seed=5)[0]
    out = O.spectrogram_regions_transform(fr, offset=20)
    assert out.shape == (4, 128, 256) and out.dtype == np.float32 and 0.0 <= out.min() and out.max() <= 1.0
    assert np.isnan(fr).any() and not np.isnan(out).any()
