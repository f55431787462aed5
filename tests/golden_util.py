"""Helpers shared by the parity tests: fixture loading and the digest rule of oracle/make_golden.py."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIG = 20000


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def summarize(t):
    f = torch.as_tensor(t).detach().double().flatten().cpu()
    ramp = torch.linspace(0.5, 1.5, f.numel(), dtype=torch.float64)
    return np.array([f.sum(), f.abs().sum(), (f * ramp).sum(), (f * f).sum()], dtype=np.float64)


def check(fix, key, value, tol=1e-5, what=""):
    """Compare ``value`` with fixture entry ``key`` (full array, or digest + head for big ones)."""
    value = torch.as_tensor(value).detach().cpu()
    if key in fix:
        want = torch.as_tensor(fix[key])
        assert tuple(value.shape) == tuple(want.shape), f"{what}{key}: shape {tuple(value.shape)} != {tuple(want.shape)}"
        err = rel_err(value, want)
        assert err <= tol, f"{what}{key}: rel err {err:.3e} > {tol}"
        return err
    assert key + "#sum" in fix, f"{key} missing from fixture"
    head = torch.as_tensor(fix[key + "#head"])
    err = rel_err(value.flatten()[: head.numel()], head)
    assert err <= tol, f"{what}{key}#head: rel err {err:.3e} > {tol}"
    got, want = summarize(value), fix[key + "#sum"]
    # abs-sum and square-sum are well conditioned; the signed sums are checked against abs-sum scale
    scale = np.array([want[1], want[1], want[1], want[3]]) + 1e-30
    derr = float(np.max(np.abs(got - want) / scale))
    assert derr <= tol, f"{what}{key}#sum: digest err {derr:.3e} > {tol}"
    return max(err, derr)
