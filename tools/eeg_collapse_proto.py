#!/usr/bin/env python3
"""fp64 check of the algebra behind the collapsed EEGNet front (conv1 -> BatchNorm1 -> depthwise electrode mix, all linear):
the [B,8,Chans,T] tensor of conv1 outputs is never formed.

  forward   v[b,fd,t] = sum_ch wd[fd,ch] x[b,ch,t]                  (mix the electrodes FIRST: 16 rows instead of 8*Chans)
            q[b,fd,t] = sum_k w1[f,k] v~[b,fd,t+k-31]                (the 64-tap convolution on the 16 mixed rows, f = fd // D)
            u = a_f q + c_f Wsum[fd],  a_f = gamma_f invstd_f,  c_f = beta_f - a_f mu_f,  Wsum[fd] = sum_ch wd[fd,ch]
  stats     mu_f = w1[f]^T S / N,  E[z^2]_f = w1[f]^T R w1[f] / N    with the input's sufficient statistics
            S[k] = sum_{b,ch,t} x~[t+k-31],  R[k,k'] = sum_{b,ch,t} x~[t+k-31] x~[t+k'-31]   (autocorrelation + edge terms)
  backward  from g = dL/du:  C[fd,ch,k] = sum_{b,t} g[b,fd,t] x~[b,ch,t+k-31],  G[fd] = sum_{b,t} g
            d wd, d gamma1, d beta1, d w1 are closed forms in (C, G, R, S)  -- see collapsed_backward().

Compares with autograd through the reference layer sequence (oracle/ref_torch.py EEGNet, M:250-275).  Run on CPU."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.manual_seed(0)
B, Ch, T, F1, D, K = 3, 5, 200, 8, 2, 64
FD = F1 * D
PADL = (K - 1) // 2
dt = torch.float64
x = torch.randn(B, Ch, T, dtype=dt) + 0.3
w1 = torch.randn(F1, K, dtype=dt, requires_grad=True)
gamma = (torch.rand(F1, dtype=dt) + 0.5).requires_grad_(True)
beta = torch.randn(F1, dtype=dt, requires_grad=True)
wd = torch.randn(FD, Ch, dtype=dt, requires_grad=True)
eps = 1e-5

# ---- reference sequence
z = F.conv2d(F.pad(x[:, None], (PADL, K - 1 - PADL)), w1[:, None, None, :])            # [B,F1,Ch,T]
mu_ref = z.mean((0, 2, 3)); var_ref = z.var((0, 2, 3), unbiased=False)
y = (z - mu_ref[None, :, None, None]) / torch.sqrt(var_ref + eps)[None, :, None, None] * gamma[None, :, None, None] + beta[None, :, None, None]
u_ref = F.conv2d(y, wd[:, None, :, None], groups=F1)[:, :, 0, :]                        # [B,FD,T]
g = torch.randn_like(u_ref)
(u_ref * g).sum().backward()

# ---- collapsed forward
with torch.no_grad():
    N = B * Ch * T
    xp = F.pad(x, (PADL, K - 1 - PADL))                                               # x~ with the conv's zero padding
    # sufficient statistics, straightforwardly (the kernel gets R from the lag sums r[d] and head / tail corrections)
    win = xp.unfold(2, T, 1)                                                          # [B,Ch,K,T]: win[..,k,t] = x~[t+k-31]
    S = win.sum((0, 1, 3))
    R = torch.einsum("bckt,bcmt->km", win, win)
    # the same R from autocorrelation lags + edge terms (what the kernel computes)
    r = torch.stack([(x[:, :, : T - d] * x[:, :, d:]).sum() for d in range(K)])
    R2 = torch.empty(K, K, dtype=dt)
    for k in range(K):
        for k2 in range(k, K):
            m, m2, d = k - PADL, k2 - PADL, k2 - k
            val = r[d].clone()
            if m > 0:
                val -= (x[:, :, :m] * x[:, :, d:d + m]).sum()
            if m2 < 0:
                j = -m2
                val -= (x[:, :, T - d - j:T - d] * x[:, :, T - j:]).sum()
            R2[k, k2] = R2[k2, k] = val
    tot = x.sum()
    S2 = torch.stack([tot - (x[:, :, :max(k - PADL, 0)].sum()) - (x[:, :, T + min(k - PADL, 0):].sum() if k - PADL < 0 else 0.0) for k in range(K)])
    print("R (lags + edges) vs direct:", float((R2 - R).abs().max() / R.abs().max()), " S:", float((S2 - S).abs().max()))
    w1d = w1.detach()
    mu = w1d @ S / N
    ez2 = torch.einsum("fk,km,fm->f", w1d, R, w1d) / N
    var = ez2 - mu * mu
    inv = 1.0 / torch.sqrt(var + eps)
    print("mean / var vs reference:", float((mu - mu_ref).abs().max()), float((var - var_ref).abs().max() / var_ref.abs().max()))
    a = gamma.detach() * inv
    c = beta.detach() - a * mu
    fidx = torch.arange(FD) // D
    wdd = wd.detach()
    v = torch.einsum("fc,bct->bft", wdd, x)                                           # mix first
    q = F.conv1d(F.pad(v, (PADL, K - 1 - PADL)), w1d[fidx][:, None, :], groups=FD)    # 16 rows, each with its filter
    u = a[fidx][None, :, None] * q + (c[fidx] * wdd.sum(1))[None, :, None]
    print("u vs reference:", float((u - u_ref).abs().max() / u_ref.abs().max()))

    # ---- collapsed backward
    def collapsed_backward(g):
        G = g.sum((0, 2))                                                             # [FD]
        Cc = torch.einsum("bft,bckt->fck", g, win)                                    # [FD,Ch,K]
        wsum = wdd.sum(1)
        zg = torch.einsum("fk,fck->fc", w1d[fidx], Cc)                                # sum_{b,t} g z[f(fd),ch]
        d_wd = a[fidx][:, None] * zg + (c[fidx] * G)[:, None]
        dbeta = torch.zeros(F1, dtype=dt).index_add_(0, fidx, wsum * G)               # sum dy
        Q = torch.zeros(F1, dtype=dt).index_add_(0, fidx, (wdd * zg).sum(1))          # sum dy z
        dgamma = inv * (Q - mu * dbeta)
        k1, k2 = dbeta / N, dgamma / N
        dyx = torch.zeros(F1, K, dtype=dt).index_add_(0, fidx, torch.einsum("fc,fck->fk", wdd, Cc))     # sum dy x~
        zx = w1d @ R                                                                  # sum z x~[..k]
        zhx = inv[:, None] * (zx - mu[:, None] * S[None, :])
        d_w1 = a[:, None] * (dyx - k1[:, None] * S[None, :] - k2[:, None] * zhx)
        return d_w1, dgamma, dbeta, d_wd

    d_w1, dgamma, dbeta, d_wd = collapsed_backward(g)
    for name, got, want in (("d conv1.weight", d_w1, w1.grad), ("d bn1.weight", dgamma, gamma.grad), ("d bn1.bias", dbeta, beta.grad),
                            ("d depthwise.weight", d_wd, wd.grad)):
        print(f"{name:20s} max rel err {float((got - want).abs().max() / want.abs().max()):.2e}")
