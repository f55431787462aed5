"""Helpers shared by the parity tests: fixture loading, the digest rule of oracle/make_golden.py and the
comparison metrics.

Metrics
  rel_err(a, b)            max|a-b| / max|b|                        -- forward values (continuous in the inputs)
  grad_close(a, b, tol..)  the same for gradients, STRICT by default.  ``floor`` floors the denominator so that a
                           gradient that is mathematically zero (e.g. d/d gamma of a BatchNorm directly followed by a
                           linear map and another BatchNorm) is compared on the scale of its neighbours, not of its
                           own rounding noise.
                           A gradient is a discontinuous function of the activations (ReLU'(0), max-pool arg-max), so
                           one pre-activation that is +1e-7 on one implementation and -1e-7 on the other changes a
                           patch of an input gradient by O(1).  Such an event is never ASSUMED here: the caller passes
                           ``flips`` = the list of OBSERVED disagreements between the implementation's stored
                           activations and the oracle's fp64 trace, each with its |z|/S margin (observed_flips()
                           below, oracle.ref_torch.activation_flips).  Only with a non-empty list is the flip-tolerant
                           branch taken: outliers must stay below ``max_outlier_frac`` of the elements and 1 % in
                           relative L2 norm, and the flips are printed with the deviation.
"""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIG = 20000
REPORT = []          # (label, max-rel-err, outlier fraction) collected for the end-of-run summary


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def _flat(t):
    return torch.as_tensor(t).detach().double().flatten().cpu()


def rel_err(a, b, floor=0.0):
    a, b = _flat(a), _flat(b)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


def grad_close(got, want, tol, label="", floor=0.0, flips=None, max_outlier_frac=0.03):
    a, b = _flat(got), _flat(want)
    assert a.shape == b.shape, f"{label}: {a.shape} vs {b.shape}"
    scale = max(float(b.abs().max()), floor, 1e-30)
    d = (a - b).abs()
    err = float(d.max() / scale)
    frac = float((d > tol * scale).double().mean())
    REPORT.append((label, err, frac))
    if err <= tol:
        return err
    where = int(d.argmax())
    assert flips, (f"{label}: rel err {err:.3e} > {tol} at flat index {where} ({frac:.2%} of elements beyond the tolerance) "
                   "and no activation flip was observed")
    l2 = float(d.norm() / max(float(b.norm()), floor * a.numel() ** 0.5, 1e-30))
    assert frac <= max_outlier_frac and l2 <= 1e-2 and err <= 0.2, (
        f"{label}: rel err {err:.3e} > {tol}, outliers {frac:.2%} (limit {max_outlier_frac:.0%}), rel L2 {l2:.2e}; "
        f"observed flips: {describe_flips(flips)}")
    print(f"[parity] {label}: {frac:.3%} of elements beyond {tol:g} (max {err:.2e} at {where}, rel L2 {l2:.2e}) "
          f"with OBSERVED activation flips: {describe_flips(flips)}")
    return err


def describe_flips(flips, limit=4):
    items = [f"{f['stage']}.conv{f['conv']}@{f['index']} |z|/S={f['margin']:.1e}" for f in flips[:limit]]
    return "; ".join(items) + (f"; ... {len(flips)} in all" if len(flips) > limit else "")


def observed_flips(oracle_mod, ref, args, keep, label=""):
    """Demonstrated activation flips of a GPU forward.  ``ref``: the oracle model in the state the forward saw (call this
    before stepping it); ``args``: the CPU inputs; ``keep``: what brainxai.ops.keep_block_activations(model) returned before
    that forward.  A disagreement whose oracle margin is NOT tiny is an error, not a flip, and fails here."""
    trace = oracle_mod.relu_pool_trace(ref, args)
    acts = {}
    for name, d in keep.items():
        if "acts" in d and name in trace:
            acts[name] = [t.detach().float().permute(0, 3, 1, 2)[:, :trace[name]["z"][k].shape[1]].cpu() for k, t in enumerate(d["acts"])]
    flips, errors = oracle_mod.activation_flips(trace, acts)
    assert not errors, f"{label}: activations disagree with the fp64 oracle beyond a tie: {describe_flips(errors)}"
    if flips:
        print(f"[parity] {label}: observed {len(flips)} activation flip(s): {describe_flips(flips)}")
    return flips


def summarize(t):
    f = _flat(t)
    ramp = torch.linspace(0.5, 1.5, f.numel(), dtype=torch.float64)
    return np.array([f.sum(), f.abs().sum(), (f * ramp).sum(), (f * f).sum()], dtype=np.float64)


def check(fix, key, value, tol=1e-5, what="", floor=0.0, robust=False, digest=True, flips=None):
    """Compare ``value`` with fixture entry ``key`` (full array, or digest + head for big ones).
    floor: absolute scale below which the reference is treated as zero; robust: gradient-style comparison."""
    value = torch.as_tensor(value).detach().cpu()
    if key in fix:
        want = torch.as_tensor(fix[key])
        assert tuple(value.shape) == tuple(want.shape), f"{what}{key}: shape {tuple(value.shape)} != {tuple(want.shape)}"
        if robust:
            return grad_close(value, want, tol, label=what + key, floor=floor, flips=flips)
        err = rel_err(value, want, floor)
        REPORT.append((what + key, err, 0.0))
        assert err <= tol, f"{what}{key}: rel err {err:.3e} > {tol}"
        return err
    assert key + "#sum" in fix, f"{key} missing from fixture"
    head = torch.as_tensor(fix[key + "#head"])
    got_head = value.flatten()[: head.numel()]
    if robust:
        err = grad_close(got_head, head, tol, label=what + key + "#head", floor=floor, flips=flips, max_outlier_frac=0.10)
    else:
        err = rel_err(got_head, head, floor)
        assert err <= tol, f"{what}{key}#head: rel err {err:.3e} > {tol}"
    if not digest:
        return err
    got, want = summarize(value), fix[key + "#sum"]
    # abs-sum and square-sum are well conditioned; the signed sums are checked against the abs-sum scale
    scale = np.array([want[1], want[1], want[1], want[3]]) + 1e-30
    derr = float(np.max(np.abs(got - want) / scale))
    dtol = max(tol, 2e-2) if (robust and flips) else tol      # an observed flip moves the digest of a gradient by O(1e-3)
    assert derr <= dtol, f"{what}{key}#sum: digest err {derr:.3e} > {dtol}"
    REPORT.append((what + key + "#sum", derr, 0.0))
    return max(err, derr)
