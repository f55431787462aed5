"""CPU oracle for the multimodal brain-pattern hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain CPU / fp32 PyTorch + numpy restatement of the reference's
algorithm for the path named in BASELINE.json.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing in the product package imports it and the
product never falls back to it.

Pinned by: ``tests/golden/*.npz`` produced by ``oracle/make_golden.py`` which
imports the reference's own classes in the build container
(/root/reference/root/src/models/models.py, XAI_Multimodality.py) and records
their outputs on seeded inputs; ``tests/test_oracle_golden.py`` replays those
fixtures through this file.

Parity status per piece
  * Block / Spectrogram_Model / EEGNet / EEGNetAttentionDeep / MultimodalModel / KLDiv /
    AdamW step / saliency / EEG stacker: pinned by reference outputs (fixtures).
  * Grad-CAM, Integrated Gradients: the reference ships NO implementation
     -> "parity unpinned" by the reference; pinned here by the
    canonical definitions applied to the *reference's* model classes when the
    fixtures were generated.

Reference anchors (paths relative to /root/reference):
  M  = root/src/models/models.py
  NB = root/jupyter_notebooks/XAI_Multimodality.py
"""
from __future__ import annotations

import math
from typing import Iterable, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

N_CLASSES = 6
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Pool schedule of the five residual stages (M:86-90).
_STAGES = ((16, "max"), (32, "avg"), (64, "max"), (128, "avg"), (256, "max"))


# --------------------------------------------------------------------------------------
# Row A -- residual stage of the spectrogram CNN (M:42-77)
# --------------------------------------------------------------------------------------
class Block(nn.Module):
    """relu(conv3x3) x3 -> 2x2 pool -> BatchNorm -> dropout -> + conv1x1(bilinear(x)).

    Follows M:47-60 (parameters) and M:62-77 (op order: pool BEFORE bn, dropout after
    bn, skip added last).  Parameter names equal the reference's so state_dicts swap.
    """

    def __init__(self, in_channels, out_channels, pool_type="max", pool_size=(2, 2), dropout_p=0.5):
        super().__init__()
        chans = (in_channels, out_channels, out_channels, out_channels)
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv2d(chans[i], chans[i + 1], 3, 1, 1))
        if pool_type not in ("max", "avg"):
            raise ValueError(f"pool_type must be 'max' or 'avg', got {pool_type!r}")
        self.pool = (nn.MaxPool2d if pool_type == "max" else nn.AvgPool2d)(kernel_size=pool_size)
        self.bn = nn.BatchNorm2d(out_channels)
        self.dropout = nn.Dropout(p=dropout_p)
        self.conv1x1 = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x):
        y = x
        for conv in (self.conv1, self.conv2, self.conv3):
            y = F.relu(conv(y))
        y = self.dropout(self.bn(self.pool(y)))
        # The reference only resamples when shapes differ (M:72) -- they always do,
        # because the channel count changes in every stage.
        skip = F.interpolate(x, size=y.shape[-2:], mode="bilinear", align_corners=False)
        return y + self.conv1x1(skip)


# --------------------------------------------------------------------------------------
# Row B -- spectrogram CNN (M:79-107)
# --------------------------------------------------------------------------------------
class Spectrogram_Model(nn.Module):
    """Five stages -> global average pool -> Linear(256, classes) -> LogSoftmax.

    ``in_channels`` is the build's extension (reference hard-codes 3, M:86); the
    benchmark's 4-plane spectrogram uses in_channels=4, equal to swapping
    ``Block(4, 16, 'max')`` into ``block1`` of the reference class.
    """

    def __init__(self, num_classes=N_CLASSES, in_channels=3):
        super().__init__()
        c_prev = in_channels
        for i, (c, kind) in enumerate(_STAGES, start=1):
            setattr(self, f"block{i}", Block(c_prev, c, kind, (2, 2)))
            c_prev = c
        self.gap = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(c_prev, num_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)

    def features(self, x):
        for i in range(1, 6):
            x = getattr(self, f"block{i}")(x)
        return x

    def forward(self, x):
        x = self.features(x)
        return self.log_softmax(self.fc(self.gap(x).flatten(1)))


# --------------------------------------------------------------------------------------
# Row C -- EEGNet over raw EEG [B,1,Chans,Samples] (M:239-289)
# --------------------------------------------------------------------------------------
class EEGNet(nn.Module):
    """temporal conv(1xK, 'same') -> BN -> depthwise (Chans x 1) -> BN -> ELU -> avgpool4
    -> dropout -> conv(1x16 'same', dense over F1*D) -> BN -> ELU -> avgpool8 -> dropout
    -> Linear -> LogSoftmax.  No conv has a bias; one Dropout / one ELU instance (M:253-255).
    """

    def __init__(self, nb_classes, Chans=37, Samples=3000, dropoutRate=0.5, kernLength=64,
                 F1=8, D=2, F2=16, norm_rate=0.25, dropoutType="Dropout"):
        super().__init__()
        self.nb_classes, self.Chans, self.Samples = nb_classes, Chans, Samples
        self.conv1 = nn.Conv2d(1, F1, (1, kernLength), padding="same", bias=False)
        self.batchnorm1 = nn.BatchNorm2d(F1)
        self.depthwiseConv = nn.Conv2d(F1, F1 * D, (Chans, 1), groups=F1, bias=False)
        self.batchnorm2 = nn.BatchNorm2d(F1 * D)
        self.activation = nn.ELU()
        self.avg_pool1 = nn.AvgPool2d((1, 4))
        self.dropout = nn.Dropout(dropoutRate) if dropoutType == "Dropout" else nn.Dropout2d(dropoutRate)
        self.separableConv = nn.Conv2d(F1 * D, F2, (1, 16), padding="same", bias=False)
        self.batchnorm3 = nn.BatchNorm2d(F2)
        self.avg_pool2 = nn.AvgPool2d((1, 8))
        self.flatten = nn.Flatten()
        self.dense = nn.Linear(F2 * (Samples // 32), nb_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)

    def stages(self, x):
        """All intermediates, in order (used by fixtures to pin pad/pool conventions)."""
        t = {}
        t["conv1"] = self.conv1(x)
        t["bn1"] = self.batchnorm1(t["conv1"])
        t["dw"] = self.depthwiseConv(t["bn1"])
        t["bn2"] = self.batchnorm2(t["dw"])
        t["pool1"] = self.dropout(self.avg_pool1(self.activation(t["bn2"])))
        t["sep"] = self.separableConv(t["pool1"])
        t["bn3"] = self.batchnorm3(t["sep"])
        t["pool2"] = self.dropout(self.avg_pool2(self.activation(t["bn3"])))
        t["out"] = self.log_softmax(self.dense(self.flatten(t["pool2"])))
        return t

    def forward(self, x):
        return self.stages(x)["out"]


# --------------------------------------------------------------------------------------
# Row C' -- deeper EEGNet with a single-head self-attention over time (M:109-235)
# --------------------------------------------------------------------------------------
class Attention(nn.Module):
    """softmax(Q K^T / sqrt(d)) V with three biased Linear maps (M:109-134); returns (output, weights)."""

    def __init__(self, input_dim, attention_dim):
        super().__init__()
        self.query = nn.Linear(input_dim, attention_dim)
        self.key = nn.Linear(input_dim, attention_dim)
        self.value = nn.Linear(input_dim, attention_dim)
        self.scale = attention_dim ** -0.5

    def forward(self, x):
        q, k, v = self.query(x), self.key(x), self.value(x)
        w = torch.softmax(q @ k.transpose(-2, -1) * self.scale, dim=-1)
        return w @ v, w


class EEGNetAttentionDeep(nn.Module):
    """EEGNet blocks 1-2 (own Dropout per block), a third block conv(1x16 'same', F2->F3) -> BN -> ELU -> avgpool8 ->
    dropout, attention over the remaining time steps (tokens = time, features = F3), then Linear(F3*L,128) ->
    Linear(128,nb_classes) with NO activation in between -> LogSoftmax (M:136-235)."""

    def __init__(self, nb_classes, Chans=37, Samples=3000, dropoutRate=0.5, kernLength=64, F1=8, D=2, F2=16, F3=32,
                 norm_rate=0.25, dropoutType="Dropout"):
        super().__init__()
        self.nb_classes, self.Chans, self.Samples = nb_classes, Chans, Samples
        drop = (lambda: nn.Dropout(dropoutRate)) if dropoutType == "Dropout" else (lambda: nn.Dropout2d(dropoutRate))
        self.conv1 = nn.Conv2d(1, F1, (1, kernLength), padding="same", bias=False)
        self.batchnorm1 = nn.BatchNorm2d(F1)
        self.depthwiseConv = nn.Conv2d(F1, F1 * D, (Chans, 1), groups=F1, bias=False)
        self.batchnorm2 = nn.BatchNorm2d(F1 * D)
        self.activation = nn.ELU()
        self.avg_pool1 = nn.AvgPool2d((1, 4))
        self.dropout1 = drop()
        self.separableConv = nn.Conv2d(F1 * D, F2, (1, 16), padding="same", bias=False)
        self.batchnorm3 = nn.BatchNorm2d(F2)
        self.avg_pool2 = nn.AvgPool2d((1, 8))
        self.dropout2 = drop()
        self.conv2 = nn.Conv2d(F2, F3, (1, 16), padding="same", bias=False)
        self.batchnorm4 = nn.BatchNorm2d(F3)
        self.avg_pool3 = nn.AvgPool2d((1, 8))
        self.dropout3 = drop()
        self.attention_layer = Attention(F3, F3)
        self.output_samples = ((Samples // 4) // 8) // 8        # what the reference finds with a dummy forward (M:180-202)
        self.flattened_size = F3 * self.output_samples
        self.flatten = nn.Flatten()
        self.dense1 = nn.Linear(self.flattened_size, 128)
        self.dense2 = nn.Linear(128, nb_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)

    def stages(self, x):
        t = {}
        u = self.batchnorm2(self.depthwiseConv(self.batchnorm1(self.conv1(x))))
        t["pool1"] = self.dropout1(self.avg_pool1(self.activation(u)))
        t["pool2"] = self.dropout2(self.avg_pool2(self.activation(self.batchnorm3(self.separableConv(t["pool1"])))))
        t["conv2"] = self.conv2(t["pool2"])
        t["bn4"] = self.batchnorm4(t["conv2"])
        t["pool3"] = self.dropout3(self.avg_pool3(self.activation(t["bn4"])))
        b, c, h, w = t["pool3"].shape
        tokens = t["pool3"].reshape(b, c, h * w).permute(0, 2, 1)          # [B, L, F3]
        o, t["attn"] = self.attention_layer(tokens)
        t["attended"] = o.permute(0, 2, 1).reshape(b, c, h, w)
        t["out"] = self.log_softmax(self.dense2(self.dense1(self.flatten(t["attended"]))))
        return t

    def forward(self, x):
        return self.stages(x)["out"]


# --------------------------------------------------------------------------------------
# Row D -- late-fusion head over the two 6-way log-prob vectors (NB:1082-1108)
# --------------------------------------------------------------------------------------
class MultimodalModel(nn.Module):
    def __init__(self, eeg_model, spectrogram_model, num_classes=N_CLASSES):
        super().__init__()
        self.eeg_model = eeg_model
        self.spectrogram_model = spectrogram_model
        width = eeg_model.dense.out_features + spectrogram_model.fc.out_features
        self.fc1 = nn.Linear(width, 128)
        self.fc2 = nn.Linear(128, num_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)

    def forward(self, eeg_data, spectrogram_data):
        z = torch.cat((self.eeg_model(eeg_data), self.spectrogram_model(spectrogram_data)), dim=1)
        return self.log_softmax(self.fc2(F.relu(self.fc1(z))))

    def forward_spectrogram(self, spectrogram_data):
        return self.spectrogram_model(spectrogram_data)


def build_multimodal(chans=19, samples=2000, in_channels=4, num_classes=N_CLASSES, dropout=0.5):
    """Benchmark model of BASELINE.md section 2: 2 025 074 parameters at the defaults."""
    eeg = EEGNet(num_classes, Chans=chans, Samples=samples, dropoutRate=dropout)
    spec = Spectrogram_Model(num_classes, in_channels=in_channels)
    if dropout != 0.5:
        for i in range(1, 6):
            getattr(spec, f"block{i}").dropout.p = dropout
    return MultimodalModel(eeg, spec, num_classes)


def set_dropout(model: nn.Module, p: float) -> None:
    for m in model.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = p


# --------------------------------------------------------------------------------------
# Row E -- one optimisation step of train_and_validate_combined (NB:1595-1607)
# --------------------------------------------------------------------------------------
def kl_div(log_probs, target, reduction="mean"):
    """nn.KLDivLoss: sum t*(log t - y) with 0*log 0 := 0; 'mean' divides by numel (NB:1989),
    'batchmean' by batch size (NB:1757)."""
    pointwise = torch.where(target > 0, target * (torch.log(target.clamp_min(1e-38)) - log_probs),
                            torch.zeros_like(log_probs))
    total = pointwise.sum()
    if reduction == "mean":
        return total / log_probs.numel()
    if reduction == "batchmean":
        return total / log_probs.shape[0]
    if reduction == "sum":
        return total
    raise ValueError(reduction)


def train_step(model, optimizer, eeg, spec, labels, criterion=None):
    """zero_grad -> forward -> KLDiv -> backward -> step; returns (loss, n_correct)."""
    criterion = criterion or nn.KLDivLoss()
    optimizer.zero_grad()
    out = model(eeg, spec)
    loss = criterion(out, labels)
    loss.backward()
    optimizer.step()
    correct = (out.argmax(1) == labels.argmax(1)).sum()
    return float(loss.detach()), int(correct)


# --------------------------------------------------------------------------------------
# Row F -- one epoch of train_and_validate_eeg_distributed (root/src/training/training_distributed.py:34-101), one rank
# --------------------------------------------------------------------------------------
def l2_penalty(model: nn.Module, weight_decay: float):
    """``reg_loss = sum(torch.sum(param ** 2) for param in model.parameters()) * weight_decay`` (DDP:52-53)."""
    return sum(torch.sum(p ** 2) for p in model.parameters()) * weight_decay


def distributed_epoch(model, train_batches, valid_batches, optimizer, criterion, weight_decay, scheduler=None, on_step=None):
    """The body of the reference's DDP epoch loop on ONE replica (DDP averaging over ranks is the identity for one rank; for N
    ranks each replica runs exactly this on its shard and gradients are averaged before ``optimizer.step()``).  Follows
    DDP:34-101 line by line: total_loss = loss + reg (:55), running sums of ``loss.item()`` / ``reg_loss.item()`` (:59-60)
    divided by the number of SAMPLES (:66-67, :95), arg-max accuracy, ``scheduler.step(valid_loss)`` (:99).
    ``on_step(i, loss, reg, model)`` is called after backward and before the optimizer step (test hook).
    Returns dict(train_loss, reg_loss, train_acc, valid_loss, valid_acc, lr)."""
    model.train()
    run_loss = run_reg = 0.0
    correct = total = 0
    for i, (data, labels) in enumerate(train_batches):
        optimizer.zero_grad()
        out = model(data)
        loss = criterion(out, labels)
        reg = l2_penalty(model, weight_decay)
        (loss + reg).backward()
        if on_step is not None:
            on_step(i, loss.detach(), reg.detach(), model)
        optimizer.step()
        run_loss += loss.item(); run_reg += reg.item()
        total += labels.size(0)
        correct += int((out.argmax(1) == labels.argmax(1)).sum())
    res = {"train_loss": run_loss / total, "reg_loss": run_reg / total, "train_acc": 100.0 * correct / total}
    model.eval()
    run_v = 0.0
    correct = total = 0
    with torch.no_grad():
        for data, labels in valid_batches:
            out = model(data)
            run_v += criterion(out, labels).item()
            total += labels.size(0)
            correct += int((out.argmax(1) == labels.argmax(1)).sum())
    res.update(valid_loss=run_v / total, valid_acc=100.0 * correct / total, lr=None)
    if scheduler is not None:
        scheduler.step(res["valid_loss"])
        res["lr"] = scheduler.get_last_lr()[0]
    return res


# --------------------------------------------------------------------------------------
# Row G -- attribution
# --------------------------------------------------------------------------------------
def _resolve(model: nn.Module, dotted: str) -> nn.Module:
    mod = model
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def _class_list(class_idx, out):
    if class_idx is None:
        return None
    if isinstance(class_idx, str):
        if class_idx != "all":
            raise ValueError(class_idx)
        return list(range(out.shape[1]))
    return [int(class_idx)]


def grad_cam(model, eeg, spec, target_layer="spectrogram_model.block5", class_idx=None,
             upsample=True, relu=True, return_parts=False):
    """Canonical Grad-CAM (Selvaraju et al.) with a forward hook on ``target_layer``.

    score  y_c = model output (a log-probability, as the reference's saliency uses, NB:3109-3111)
    w[b,k]   = mean_{h,w} d y_c / d A[b,k,h,w]
    cam[b]   = ReLU( sum_k w[b,k] * A[b,k] )     -> optional bilinear upsample to the input H x W

    class_idx None -> each sample's argmax class; int -> that class; 'all' -> every class
    (output gains a class axis: [B, n_classes, H, W]).
    """
    was_training = model.training
    model.eval()
    grabbed = {}
    handle = _resolve(model, target_layer).register_forward_hook(
        lambda _m, _i, o: grabbed.__setitem__("A", o))
    try:
        out = model(eeg, spec)
    finally:
        handle.remove()
    A = grabbed["A"]
    classes = _class_list(class_idx, out)
    maps, raws, weights = [], [], []
    for c in ([None] if classes is None else classes):
        if c is None:
            score = out.gather(1, out.argmax(1, keepdim=True)).sum()
        else:
            score = out[:, c].sum()
        (G,) = torch.autograd.grad(score, A, retain_graph=True)
        w = G.mean(dim=(2, 3), keepdim=True)
        raw = (w * A).sum(dim=1)
        cam = raw.clamp_min(0) if relu else raw
        if upsample:
            cam = F.interpolate(cam[:, None], size=spec.shape[-2:], mode="bilinear",
                                align_corners=False)[:, 0]
        maps.append(cam.detach()); raws.append(raw.detach()); weights.append(w.detach()[:, :, 0, 0])
    model.train(was_training)
    stack = (lambda xs: xs[0]) if classes is None or not isinstance(class_idx, str) else (lambda xs: torch.stack(xs, 1))
    if return_parts:
        return stack(maps), stack(raws), stack(weights), A.detach(), out.detach()
    return stack(maps)


def saliency(model, eeg, spec, reference_quirk=False):
    """|d max-logprob / d input| maps (NB:3101-3129).

    Returns (eeg_sal [B,Chans,S], spec_sal [B,H,W]) with spec reduced by max over channels.
    ``reference_quirk=True`` reproduces the reference's double accumulation: its second
    backward() adds to the un-zeroed input gradient, so its spectrogram map is exactly
    2x the true one; the reference handles batch size 1 only.
    """
    was_training = model.training
    model.eval()
    eeg = eeg.detach().clone().requires_grad_(True)
    spec = spec.detach().clone().requires_grad_(True)
    out = model(eeg, spec)
    score = out.gather(1, out.argmax(1, keepdim=True)).sum()
    g_eeg, g_spec = torch.autograd.grad(score, (eeg, spec))
    model.train(was_training)
    factor = 2.0 if reference_quirk else 1.0
    return g_eeg.abs()[:, 0], factor * g_spec.abs().amax(dim=1)


def ig_nodes(n_steps=50):
    """Captum's default rule ('gausslegendre'): alphas and step sizes on [0,1]."""
    x, w = np.polynomial.legendre.leggauss(n_steps)
    return 0.5 * (1.0 + x), 0.5 * w


def integrated_gradients(model, inputs, baselines=None, target=None, n_steps=50):
    """(x - x') * sum_k w_k grad F_target(x' + a_k (x - x')); Captum default semantics.

    inputs = (eeg, spec); target None -> argmax class of the un-interpolated input.
    """
    was_training = model.training
    model.eval()
    eeg, spec = inputs
    b_eeg, b_spec = baselines if baselines is not None else (torch.zeros_like(eeg), torch.zeros_like(spec))
    with torch.no_grad():
        tgt = model(eeg, spec).argmax(1) if target is None else torch.full((eeg.shape[0],), int(target))
    alphas, steps = ig_nodes(n_steps)
    acc_e, acc_s = torch.zeros_like(eeg), torch.zeros_like(spec)
    for a, s in zip(alphas, steps):
        xe = (b_eeg + float(a) * (eeg - b_eeg)).requires_grad_(True)
        xs = (b_spec + float(a) * (spec - b_spec)).requires_grad_(True)
        out = model(xe, xs)
        score = out.gather(1, tgt[:, None]).sum()
        ge, gs = torch.autograd.grad(score, (xe, xs))
        acc_e += float(s) * ge
        acc_s += float(s) * gs
    model.train(was_training)
    return acc_e * (eeg - b_eeg), acc_s * (spec - b_spec)


# --------------------------------------------------------------------------------------
# Row H -- EEG stacker: raw [L, C] window -> [19, L/5]  (dataset.py:73-104,125-131,213-228)
# --------------------------------------------------------------------------------------
def butter_lowpass_coeffs(cutoff_freq=20.0, sampling_rate=200.0, order=4):
    from scipy.signal import butter
    return butter(order, cutoff_freq / (0.5 * sampling_rate), btype="low", analog=False)


def eeg_transform(x: np.ndarray, channels: Optional[Sequence[int]] = None, downsample=5) -> np.ndarray:
    """select -> clip(+-1024) -> NaN->0 -> /32 -> 4th-order Butterworth low-pass 20 Hz
    (scipy lfilter along time, float64) -> every 5th sample.  Returns float32 [L/5, C]."""
    from scipy.signal import lfilter
    if channels is not None:
        x = x[:, list(channels)]
    y = np.nan_to_num(np.clip(x, -1024, 1024), nan=0) / 32.0
    b, a = butter_lowpass_coeffs()
    y = lfilter(b, a, y, axis=0)
    return y[::downsample, :].astype(np.float32)


def stack_eeg_batch(raw: np.ndarray) -> torch.Tensor:
    """[B, L, C] raw -> EEGNet input [B, 1, C, L/5] (EEGDataset.__getitem__ permute + unsqueeze)."""
    out = np.stack([eeg_transform(r).T for r in raw])
    return torch.from_numpy(out)[:, None]


# --------------------------------------------------------------------------------------
# 8(f) rank 3 -- native-pipeline EEG montage stacker: raw [L, 20] frame -> [1, 37, 3000]
# (reference XAI_Multimodality.py: CombinedDataset.process_eeg :1148-1164 and its helpers :1211-1276, CFG :115-241)
# --------------------------------------------------------------------------------------
EEG_COLUMNS = ["Fp1", "F3", "C3", "P3", "F7", "T3", "T5", "O1", "Fz", "Cz", "Pz", "Fp2", "F4", "C4", "P4", "F8", "T4", "T6", "O2", "EKG"]
MAP_FEATURES = [("Fp1", "F7"), ("F7", "T3"), ("T3", "T5"), ("T5", "O1"), ("Fp1", "F3"), ("F3", "C3"), ("C3", "P3"), ("P3", "O1"),
                ("Fp2", "F8"), ("F8", "T4"), ("T4", "T6"), ("T6", "O2"), ("Fp2", "F4"), ("F4", "C4"), ("C4", "P4"), ("P4", "O2"),
                ("Fz", "Cz"), ("Cz", "Pz")]


def montage_coeffs(low=0.5, high=20.0, fs=200.0):
    """(b5, a5), (b6, a6): Butterworth band-pass designs of order 5 (first pass) and 6 (denoise pass) -- NB:1245-1253,1264."""
    from scipy.signal import butter
    nyq = 0.5 * fs
    return butter(5, [low / nyq, high / nyq], btype="band"), butter(6, [low / nyq, high / nyq], btype="band")


def montage_rows():
    """(a, b) source rows of the 37 selected output rows; b = -1 for a plain channel.  Reproduces the reference's
    off-by-one (NB:1126-1127,1271-1276): the bipolar rows are appended AFTER all 20 raw rows (EKG included) but are
    selected from index 19 = len(feature_to_index), so the output is 19 EEG channels + EKG + the first 17 differences."""
    idx = {n: i for i, n in enumerate(EEG_COLUMNS[:19])}
    rows = [(i, -1) for i in range(19)] + [(19, -1)]
    rows += [(idx[a], idx[b]) for a, b in MAP_FEATURES[:17]]
    return rows


def montage_transform(frame: np.ndarray, fixed_length=3000) -> np.ndarray:
    """frame: float32 [L, 20] (columns in EEG_COLUMNS order, as ``eeg.values``).  Returns float32 [1, 37, fixed_length].
    band-pass(order 5, lfilter) -> NaN -> row nanmean -> append 18 bipolar differences -> band-pass(order 6) ->
    4-tap forward mean (np.roll over the FLATTENED array, as the reference writes it) -> every 4th column of [0, L-1) ->
    per-row z-score (population std, eps 1e-6) -> select 19 + rows 19..36 -> zero-pad / truncate."""
    from scipy.signal import lfilter
    (b5, a5), (b6, a6) = montage_coeffs()
    waves = frame.T
    waves = lfilter(b5, a5, waves, axis=1)
    data = waves[~np.isnan(waves).all(axis=1)]                                   # handle_nan (NB:1211-1221)
    if data.size == 0:
        raise ValueError("every row is NaN")                                     # reference substitutes a mis-sized zero array
    where_nan = np.isnan(data)
    mean_values = np.nanmean(data, axis=1, keepdims=True) if where_nan.any() else np.zeros((data.shape[0], 1))
    mean_values[np.isnan(mean_values)] = 0
    data[where_nan] = np.take(mean_values, np.where(where_nan)[0])
    idx = {n: i for i, n in enumerate(EEG_COLUMNS[:19])}
    diff = np.zeros((len(MAP_FEATURES), data.shape[1]))
    for i, (fa, fb) in enumerate(MAP_FEATURES):
        diff[i] = data[idx[fa]] - data[idx[fb]]
    data = np.vstack((data, diff))
    y = lfilter(b6, a6, data, axis=1)                                            # denoise_filter (NB:1263-1267)
    y = (y + np.roll(y, -1) + np.roll(y, -2) + np.roll(y, -3)) / 4
    y = y[:, 0:-1:4]
    y = (y - np.mean(y, axis=1, keepdims=True)) / (np.std(y, axis=1, keepdims=True) + 1e-6)
    sel = list(range(19)) + list(range(19, 19 + len(MAP_FEATURES)))
    y = y[sel]
    if y.shape[1] < fixed_length:
        y = np.hstack((y, np.zeros((y.shape[0], fixed_length - y.shape[1]))))
    else:
        y = y[:, :fixed_length]
    return y[np.newaxis].astype(np.float32)


def synthetic_frames(batch=2, length=10000, seed=7, nan_rate=1e-4):
    """raw frames [B, L, 20] like the Kaggle parquet (microvolt scale, slow drift + 10 Hz rhythm + noise, a few NaNs
    that never sit at t = 0, so no row is NaN from the start)."""
    g = np.random.default_rng(seed)
    t = np.arange(length)[None, :, None] / 200.0
    x = 40 * g.standard_normal((batch, length, 20)) + 60 * np.sin(2 * np.pi * 10 * t + g.uniform(0, 6, (batch, 1, 20))) + 200 * g.standard_normal((batch, 1, 20))
    x = x.astype(np.float32)
    m = g.random((batch, length, 20)) < nan_rate
    m[:, :64] = False
    x[m] = np.nan
    return x


# --------------------------------------------------------------------------------------
# 8(f) rank 2 -- native-pipeline spectrogram pre-processing: parquet values [Trows, 400] -> [3, 400, 300]
# (reference XAI_Multimodality.py: CombinedDataset.process_spectrogram :1166-1204 with helpers :1211-1243,1288-1307)
# skimage.transform.resize (:1305-1307) is called with the array's own shape (400, 300): zoom factor 1, anti-aliasing
# sigma 0 -> the identity.  scikit-image is not installed here, so that step is NOT pinned by execution.
# --------------------------------------------------------------------------------------
def notch_coeffs(freq=60.0, fs=200.0, quality=30.0):
    from scipy.signal import iirnotch, lfilter_zi
    b, a = iirnotch(freq, quality, fs)
    return b, a, lfilter_zi(b, a)


def gaussian_weights(sigma=1.0, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius) with radius = int(truncate * sigma + 0.5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def spectrogram_transform(raw: np.ndarray, offset=None, image_size=(400, 300)) -> np.ndarray:
    """raw: float [Trows, C] = ``load_train_spectr_frame(id).to_numpy()`` without the time column.  float32 [3, 400, 300].
    (offset // 2 column window of 300, zero padded) -> transpose -> pad/truncate to 400 x 300 -> drop all-NaN rows, NaN ->
    row nanmean -> subtract column means -> 60 Hz notch filtfilt along axis 0 -> gaussian sigma 1 -> NaN -> nanmean,
    min-max to [0, 1) -> resize to its own shape (identity) -> 3 identical channels."""
    from scipy.ndimage import gaussian_filter
    from scipy.signal import filtfilt
    x = np.asarray(raw, dtype=np.float64) if raw.dtype != np.float32 else raw
    if offset is not None:
        o = offset // 2
        basic = x[:, o:o + 300]
        basic = np.pad(basic, ((0, 0), (0, max(0, 300 - basic.shape[1]))), mode="constant")
    else:
        basic = x
    s = basic.T
    rows, cols = image_size
    s = np.vstack((s, np.zeros((rows - s.shape[0], s.shape[1])))) if s.shape[0] < rows else s[:rows, :]
    s = np.hstack((s, np.zeros((s.shape[0], cols - s.shape[1])))) if s.shape[1] < cols else s[:, :cols]
    s = s[~np.isnan(s).all(axis=1)]
    if s.shape[0] != rows:
        raise ValueError("a row is entirely NaN: the reference drops it and really resamples afterwards (needs scikit-image)")
    where_nan = np.isnan(s)
    if where_nan.any():
        mv = np.nanmean(s, axis=1, keepdims=True)
        mv[np.isnan(mv)] = 0
        s = s.copy()
        s[where_nan] = np.take(mv, np.where(where_nan)[0])
    s = s - np.mean(s, axis=0)
    b, a, _ = notch_coeffs()
    s = filtfilt(b, a, s, axis=0)
    s = gaussian_filter(s, sigma=1.0)
    s = np.nan_to_num(s, nan=np.nanmean(s))
    s = (s - np.min(s)) / (np.max(s) - np.min(s) + 1e-6)
    return np.tile(s[..., None], (1, 1, 3)).astype(np.float32).transpose(2, 0, 1)


# --------------------------------------------------------------------------------------
# Row H, spectrogram half (benchmark variant) -- four region planes [4, 128, 256] from one parquet frame
# The reference has no 4-plane pipeline (its Spectrogram_Model takes one 400 x 300 image tiled to 3 channels);
# BASELINE.json's [64, 4, 128, 256] input is the four 100-bin regions of a Kaggle HMS spectrogram (LL, RL, RP, LP: 400 columns)
# stacked as channels.  The stacker composes the reference's OWN helpers on that layout:
#   window of 300 time rows from offset // 2 (zero padded, process_spectrogram :1178-1183) -> transpose to [400, 300] ->
#   normalize_signal (data_utils.py:133-136: NaN -> nanmean of the sample, min-max with 1e-6) ->
#   per region resample_spectrogram (data_utils.py:145-147: skimage.transform.resize(mode='reflect', anti_aliasing=True)) to 128 x 256.
# scikit-image (pinned 0.24.0) is not installed: its resize for this case IS two scipy.ndimage calls (skimage/transform/
# _warps.py: gaussian_filter with sigma = max(0, (s - 1) / 2) per axis, mode 'mirror' for 'reflect', then
# ndi.zoom(order=1, mode='mirror', grid_mode=True), output clipped to the input's range), restated here with scipy itself.
# Parity status: unpinned by the reference (no such path, no vectors); pinned to scipy's published behaviour.
# --------------------------------------------------------------------------------------
def skimage_resize(image: np.ndarray, output_shape, anti_aliasing=True) -> np.ndarray:
    """skimage.transform.resize(image, output_shape, order=1, mode='reflect', anti_aliasing=..., clip=True) for 2-D float input."""
    from scipy import ndimage as ndi
    image = np.asarray(image, dtype=np.float64)
    factors = np.divide(image.shape, output_shape)
    if anti_aliasing:
        sigma = np.maximum(0, (factors - 1) / 2)
        if np.any(sigma > 0):
            image = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    out = ndi.zoom(image, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    return np.clip(out, image.min(), image.max())


def resize_gaussian_weights(n_in, n_out, truncate=4.0):
    """1-D anti-aliasing kernel skimage.resize applies along an axis that shrinks from n_in to n_out ([1.0] when it does not)."""
    sigma = max(0.0, (n_in / n_out - 1.0) / 2.0)
    radius = int(truncate * sigma + 0.5)
    if radius == 0:
        return np.ones(1)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def spectrogram_regions_transform(raw: np.ndarray, offset=None, out_hw=(128, 256), window=300, regions=4) -> np.ndarray:
    """raw: float [Trows, 400] parquet values (NaNs allowed).  float32 [regions, out_h, out_w]."""
    raw = np.asarray(raw, dtype=np.float64)
    o = 0 if offset is None else int(offset) // 2
    win = raw[o:o + window]
    if win.shape[0] < window:
        win = np.vstack((win, np.zeros((window - win.shape[0], raw.shape[1]))))
    sig = win.T.copy()                                           # [400, 300]: frequency bins x time
    sig = np.nan_to_num(sig, nan=np.nanmean(sig))                # normalize_signal (data_utils.py:133-136)
    sig = (sig - np.min(sig)) / (np.max(sig) - np.min(sig) + 1e-6)
    per = sig.shape[0] // regions
    return np.stack([skimage_resize(sig[r * per:(r + 1) * per], out_hw) for r in range(regions)]).astype(np.float32)


def synthetic_spectrogram_frames(batch=2, trows=320, seed=5, nan_rate=2e-3):
    """parquet-like spectrogram values [B, Trows, 400]: positive, heavy-tailed power values with a few NaNs (never a whole column)."""
    g = np.random.default_rng(seed)
    x = np.exp(g.standard_normal((batch, trows, 400)) * 1.5 + np.linspace(2, -2, 400)[None, None, :]).astype(np.float32)
    m = g.random(x.shape) < nan_rate
    x[m] = np.nan
    return x


# --------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY.md section 8(d) -- shared by tests, smoke and bench
# --------------------------------------------------------------------------------------
def synthetic_batch(batch=64, in_channels=4, height=128, width=256, chans=19, raw_len=10000, seed=42,
                    stacked=True):
    g = torch.Generator().manual_seed(seed)
    spec = torch.rand(batch, in_channels, height, width, generator=g)
    raw = torch.randn(batch, raw_len, chans, generator=g) * 100.0
    flat = raw.view(-1)
    n = flat.numel()
    k = max(1, n // 1000)
    idx = torch.randint(0, n, (2 * k,), generator=g)
    flat[idx[:k]] = float("nan")
    flat[idx[k:]] *= 50.0
    labels = torch.softmax(torch.randn(batch, N_CLASSES, generator=g), dim=1)
    eeg = stack_eeg_batch(raw.numpy()) if stacked else None
    return {"spec": spec, "raw_eeg": raw, "eeg": eeg, "labels": labels}


# --------------------------------------------------------------------------------------
# Deterministic, construction-order-independent test weights (fixtures carry no weights)
# --------------------------------------------------------------------------------------
def fill_params(model: nn.Module, seed: int = 42) -> nn.Module:
    """Overwrite every parameter/buffer from a generator keyed by (seed, tensor name).

    Conv/linear weights ~ U(+-sqrt(6/fan_in)) (keeps activations O(1) through ReLU chains),
    biases ~ U(+-0.1), BN gamma ~ U(0.5,1.5), beta/running_mean ~ U(+-0.2),
    running_var ~ U(0.5,1.5), counters 0.  Works on the reference's classes and on the
    build's classes alike because the tensor *names* are identical.
    """
    import zlib
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                t.zero_()
            elif leaf == "running_var":
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            elif leaf == "running_mean":
                t.copy_(torch.rand(t.shape, generator=g) * 0.4 - 0.2)
            elif t.dim() == 1 and leaf == "weight":          # BatchNorm gamma
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            elif t.dim() == 1:                                # any bias / BN beta
                t.copy_(torch.rand(t.shape, generator=g) * 0.2 - 0.1)
            else:
                fan_in = t[0].numel()
                bound = math.sqrt(6.0 / fan_in)
                t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * bound)
    return model


def seeded(shape, seed, kind="rand", scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.rand(*shape, generator=g) if kind == "rand" else torch.randn(*shape, generator=g)
    return t * scale


def summarize(t: torch.Tensor) -> np.ndarray:
    """Order-sensitive 4-number digest of a tensor + used with a head slice in fixtures."""
    f = t.detach().double().flatten()
    ramp = torch.linspace(0.5, 1.5, f.numel(), dtype=torch.float64)
    return np.array([f.sum(), f.abs().sum(), (f * ramp).sum(), (f * f).sum()], dtype=np.float64)


def expected_gradients(model, x, background, nsamples=200, seed=0):
    """SHAP GradientExplainer's estimator (reference call sites XAI_Multimodality.py:2283-2290; the arithmetic lives in the
    un-vendored `shap` package -> parity unpinned by the reference, canonical definition used):
        phi_c(x) = E_{b ~ background, a ~ U(0,1)} [ (x - b) * d f_c / d x (b + a (x - b)) ]
    Single-input model (the reference explains `multimodal_model.eeg_model`).  The (baseline index, alpha) draws come from
    numpy's default_rng(seed) in sample-major order, so any implementation can reproduce them.  Returns [B, n_classes, ...]."""
    was_training = model.training
    model.eval()
    rng = np.random.default_rng(seed)
    B = x.shape[0]
    with torch.no_grad():
        n_cls = model(x[:1]).shape[1]
    out = torch.zeros(B, n_cls, *x.shape[1:])
    for i in range(B):
        idx = rng.integers(0, background.shape[0], size=nsamples)
        alpha = torch.from_numpy(rng.random(nsamples).astype(np.float32))
        base = background[idx]
        diff = x[i:i + 1] - base
        xi = (base + alpha.view(-1, *([1] * (x.dim() - 1))) * diff).requires_grad_(True)
        y = model(xi)
        for c in range(n_cls):
            (g,) = torch.autograd.grad(y[:, c].sum(), xi, retain_graph=True)
            out[i, c] = (g * diff).mean(0)
    model.train(was_training)
    return out


# --------------------------------------------------------------------------------------
# Discontinuity bookkeeping for gradient comparisons (test tooling, no reference counterpart)
# --------------------------------------------------------------------------------------
# A gradient is a discontinuous function of the activations: ReLU'(z) jumps at z = 0 and a 2x2 max-pool routes its
# gradient to the arg-max.  Two fp32 implementations of the same convolution differ by ~1e-7 * S in a pre-activation
# (S = sum of |terms|), so a z that is that close to zero -- or two window entries that close to each other -- can land on
# different sides.  These helpers make such an event an OBSERVED fact instead of an assumption: the fp64 run of the
# oracle gives every pre-activation with its summation scale, and `activation_flips` lists the positions where another
# implementation's post-ReLU activations disagree with it, each with its |z| / S margin.
def relu_pool_trace(model: nn.Module, args, dtype=torch.float64):
    """Run a deep copy of ``model`` (same mode, same buffers) in ``dtype`` on ``args`` and record for every residual stage
    (anything with conv1..conv3 + pool, i.e. Block here and in the reference) the pre-ReLU outputs ``z`` of its three
    convolutions and their summation scales ``S = conv(|x|, |w|) + |b|``.  Returns {stage name: {"z": [z1,z2,z3],
    "S": [S1,S2,S3], "pool": "max"|"avg"}} (NCHW tensors)."""
    import copy
    twin = copy.deepcopy(model).to(dtype)
    twin.train(model.training)
    trace, hooks = {}, []
    for name, mod in twin.named_modules():
        if all(hasattr(mod, f"conv{k}") for k in (1, 2, 3)) and hasattr(mod, "pool"):
            ent = {"z": [None] * 3, "S": [None] * 3, "pool": "max" if isinstance(mod.pool, nn.MaxPool2d) else "avg"}
            trace[name] = ent
            for k in range(3):
                def hook(m, inp, out, ent=ent, k=k):
                    ent["z"][k] = out.detach()
                    ent["S"][k] = F.conv2d(inp[0].detach().abs(), m.weight.detach().abs(), m.bias.detach().abs(), padding=1)
                hooks.append(getattr(mod, f"conv{k + 1}").register_forward_hook(hook))
    with torch.no_grad():
        twin(*[a.to(dtype) if torch.is_floating_point(a) else a for a in args])
    for h in hooks:
        h.remove()
    return trace


def activation_flips(trace, acts, tie=1e-4):
    """Compare another implementation's post-ReLU activations with an fp64 ``relu_pool_trace``.

    acts: {stage name: [y1, y2, y3]} NCHW tensors (what the implementation stored after each conv+ReLU).
    Returns (flips, errors): ``flips`` = list of dicts {stage, conv|'pool', index, margin} for every position where the
    sign pattern (y > 0 vs z > 0) or a max-pool arg-max differs AND the oracle's margin |z|/S (or top-2 gap / S) is below
    ``tie`` -- a demonstrated tie flip; ``errors`` = the same for disagreements with a LARGER margin, which no rounding
    argument explains (a test must fail on those)."""
    flips, errors = [], []
    for name, ent in trace.items():
        if name not in acts:
            continue
        for k in range(3):
            z, S = ent["z"][k], ent["S"][k]
            y = torch.as_tensor(acts[name][k]).to(z.dtype)
            bad = (y > 0) != (z > 0)
            for idx in bad.nonzero().tolist():
                m = float(z[tuple(idx)].abs() / S[tuple(idx)].clamp_min(1e-300))
                (flips if m < tie else errors).append({"stage": name, "conv": k + 1, "index": idx, "margin": m})
        if ent["pool"] == "max":
            z3, S3 = F.relu(ent["z"][2]), ent["S"][2]
            y3 = torch.as_tensor(acts[name][2]).to(z3.dtype)
            Hc, Wc = z3.shape[2] // 2 * 2, z3.shape[3] // 2 * 2

            def windows(t):
                t = t[:, :, :Hc, :Wc]
                return t.reshape(t.shape[0], t.shape[1], Hc // 2, 2, Wc // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(*t.shape[:2], Hc // 2, Wc // 2, 4)
            wz, wy, wS = windows(z3), windows(y3), windows(S3)
            top2 = wz.topk(2, dim=-1).values
            # first-maximum rule on both sides (PyTorch's max_pool2d backward routes to the first maximum in window order)
            bad = (wz.argmax(-1) != wy.argmax(-1)) & (top2[..., 0] > 0)
            for idx in bad.nonzero().tolist():
                t = tuple(idx)
                m = float((top2[t][0] - top2[t][1]) / wS[t].max().clamp_min(1e-300))
                (flips if m < tie else errors).append({"stage": name, "conv": "pool", "index": idx, "margin": m})
    return flips, errors


class _StoreAs(torch.autograd.Function):
    """Emulates a tensor that lives in memory in a narrower type: the value is rounded to ``store`` on the way forward and the
    gradient that flows back through the same point is rounded too (an activation and its gradient are both stored tensors)."""

    @staticmethod
    def forward(ctx, x, store):
        ctx.store = store
        return x.to(store).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.store).to(g.dtype), None


class _StoreGradAs(torch.autograd.Function):
    """Identity forward; the gradient flowing back is rounded to ``store`` (a gradient tensor that is written to memory in the
    narrow type before it is added to another one)."""

    @staticmethod
    def forward(ctx, x, store):
        ctx.store = store
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.store).to(g.dtype), None


def decision_matched_twin(model: nn.Module, acts, dtype=torch.float64, storage=None):
    """A deep copy of ``model`` in ``dtype`` whose residual stages take their DISCRETE decisions from another implementation's
    stored activations ``acts`` ({stage name: [y1, y2, y3]} NCHW, post-ReLU): ReLU passes exactly where y_k > 0 there, and a 2x2
    max-pool picks the window element that is the (first) maximum of y3 there.  Everything continuous is recomputed in ``dtype``.
    With the decisions pinned, outputs and gradients are smooth functions of the arithmetic again, so they can be compared
    strictly; whether the pinned decisions are legitimate is a separate, explicit check (activation_flips: every disagreement with
    the exact forward must be a demonstrated tie).  With no disagreement this twin computes exactly what the plain model does.

    ``storage=torch.bfloat16`` additionally restates the product's bf16 STORAGE mode on the reference's algorithm: every tensor
    the HIP path keeps in memory as bf16 -- the converted input, each convolution's ReLU'd output, the pooled map (an average of
    four; a maximum is already one of them), each stage's output, the EEG branch's temporal-convolution output -- and the gradient
    arriving at each of those points is rounded to bf16 (round to nearest even, as v_cvt_pk_bf16_f32 does); sums, statistics,
    parameters and parameter gradients stay in ``dtype`` (the kernels accumulate in fp32).

    ``twin.decision_log`` (filled by each forward) lists, per convolution / max-pool, how many pinned decisions the twin's own
    arithmetic would have taken the other way and the largest margin |z|/S (or window gap / S) among them: the evidence that
    those decisions are ties at the resolution of the arithmetic."""
    import copy
    import types
    twin = copy.deepcopy(model).to(dtype)
    twin.train(model.training)
    st = (lambda t: _StoreAs.apply(t, storage)) if storage is not None else (lambda t: t)
    # the skip path's input gradient (transposed bilinear of W1x1^T dOut) is a stored tensor of its own on the HIP path: it is
    # rounded once before the main path's data gradient is added to it (and the sum is rounded again)
    stg = (lambda t: _StoreGradAs.apply(t, storage)) if storage is not None else (lambda t: t)
    log = []
    twin.decision_log = log          # filled by forward(): pinned decisions this arithmetic would have taken differently, with margins

    def windows(t, Hc, Wc):
        t = t[:, :, :Hc, :Wc]
        return t.reshape(t.shape[0], t.shape[1], Hc // 2, 2, Wc // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(*t.shape[:2], Hc // 2, Wc // 2, 4)

    first = True
    for name, mod in twin.named_modules():
        if name not in acts or not all(hasattr(mod, f"conv{k}") for k in (1, 2, 3)):
            continue
        theirs = [torch.as_tensor(a).to(dtype) for a in acts[name]]

        def forward(self, x, theirs=theirs, is_first=first, name=name):
            if is_first:
                x = st(x)                                   # the fp32 input batch is converted to the storage type once
            y = x
            for k, conv in enumerate((self.conv1, self.conv2, self.conv3)):
                z = conv(y)
                mask = theirs[k] > 0
                with torch.no_grad():                       # how far from a tie is every pinned decision that this arithmetic would not take?
                    bad = (z > 0) != mask
                    if bool(bad.any()):
                        S = F.conv2d(y.abs(), conv.weight.abs(), conv.bias.abs(), padding=1)
                        log.append({"stage": name, "conv": k + 1, "count": int(bad.sum()), "margin": float((z.abs() / S.clamp_min(1e-300))[bad].max())})
                y = st(z * mask.to(z.dtype))
            if isinstance(self.pool, nn.MaxPool2d):
                Hc, Wc = y.shape[2] // 2 * 2, y.shape[3] // 2 * 2
                idx = windows(theirs[2], Hc, Wc).argmax(-1, keepdim=True)
                wy = windows(y, Hc, Wc)
                with torch.no_grad():
                    gap = wy.max(-1).values - wy.gather(-1, idx).squeeze(-1)
                    if bool((gap > 0).any()):
                        S = windows(F.conv2d(theirs[1].abs(), self.conv3.weight.abs(), self.conv3.bias.abs(), padding=1), Hc, Wc).max(-1).values   # scale of conv3's sums
                        log.append({"stage": name, "conv": "pool", "count": int((gap > 0).sum()), "margin": float((gap / S.clamp_min(1e-300)).max())})
                y = wy.gather(-1, idx).squeeze(-1)
            else:
                y = st(self.pool(y))
            y = self.dropout(self.bn(y))
            skip = F.interpolate(stg(x), size=y.shape[-2:], mode="bilinear", align_corners=False)
            return st(y + self.conv1x1(skip))
        mod.forward = types.MethodType(forward, mod)
        first = False
    if storage is not None:
        for name, mod in twin.named_modules():              # EEGNet: only the temporal convolution's output is a narrow tensor
            if hasattr(mod, "depthwiseConv") and hasattr(mod, "conv1") and hasattr(mod, "batchnorm1"):
                conv1 = mod.conv1
                orig = conv1.forward
                conv1.forward = (lambda x, orig=orig: st(orig(x)))
    return twin


def conditioning(model: nn.Module, args, loss_of, dtype=torch.float64):
    """How well-posed a gradient comparison at fp32 is: max over parameters of the relative L2 distance between the fp32
    and the ``dtype`` gradients of ``loss_of(model(*args))`` (both on deep copies; ``model`` itself is untouched)."""
    import copy
    grads = []
    for dt in (torch.float32, dtype):
        twin = copy.deepcopy(model).to(dt)
        twin.train(model.training)
        twin.zero_grad()
        loss_of(twin(*[a.to(dt) if torch.is_floating_point(a) else a for a in args])).backward()
        grads.append({n: p.grad.detach().double() for n, p in twin.named_parameters() if p.grad is not None})
    gmax = max(float(g.norm()) for g in grads[1].values())
    worst, where = 0.0, None
    for n, g in grads[1].items():
        d = float((grads[0][n] - g).norm() / max(float(g.norm()), 1e-3 * gmax, 1e-300))
        if d > worst:
            worst, where = d, n
    return worst, where
