"""GPU EEG stacker: raw windows [B, L, C_raw] -> EEGNet input [B, 1, 19, L/5].

Replaces ``_EEGTransformer.transform`` + ``EEGDataset.__getitem__`` of the reference
(root/src/data/dataset.py:73-104,125-131,213-228): select the 19 canonical channels, clip to +-1024,
NaN -> 0, divide by 32, 4th-order Butterworth low-pass at 20 Hz (fs 200 Hz, ``lfilter`` semantics, fp64
state), keep every 5th sample, transpose to channels-first.  The filter design (scipy.signal.butter, as
the reference calls it) happens once on the host; the per-sample work is one kernel launch per batch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .ops import _p, _stream

# reference root/src/utils/cfg_utils.py:92-95
CHANNEL_FEATS = ["Fp1", "F3", "C3", "P3", "F7", "T3", "T5", "O1", "Fz", "Cz", "Pz",
                 "Fp2", "F4", "C4", "P4", "F8", "T4", "T6", "O2"]


class EEGStacker:
    def __init__(self, channel_index=None, cutoff_freq=20.0, sampling_rate=200.0, order=4, downsample=5,
                 clip=1024.0, scale=32.0):
        from scipy.signal import butter
        b, a = butter(order, cutoff_freq / (0.5 * sampling_rate), btype="low", analog=False)
        self.b = np.ascontiguousarray(b, dtype=np.float64)
        self.a = np.ascontiguousarray(a, dtype=np.float64)
        self.order, self.downsample, self.clip, self.scale = int(order), int(downsample), float(clip), float(scale)
        self.channel_index = None if channel_index is None else [int(i) for i in channel_index]
        self._idx_dev = {}

    def __call__(self, raw: torch.Tensor) -> torch.Tensor:
        if not raw.is_cuda:
            raise RuntimeError("brainxai.EEGStacker: raw EEG must be a CUDA tensor; there is no CPU path")
        if raw.dim() != 3:
            raise RuntimeError(f"expected raw EEG [B, L, C], got {tuple(raw.shape)}")
        raw = raw.contiguous().float()
        B, Lr, Craw = raw.shape
        idx = None
        n_out = Craw
        if self.channel_index is not None:
            if max(self.channel_index) >= Craw:
                raise RuntimeError("channel index exceeds the raw channel count")
            key = raw.device.index
            if key not in self._idx_dev:
                self._idx_dev[key] = torch.tensor(self.channel_index, dtype=torch.int32, device=raw.device)
            idx = self._idx_dev[key]
            n_out = len(self.channel_index)
        if B == 0:
            return torch.empty(0, 1, n_out, (Lr + self.downsample - 1) // self.downsample, device=raw.device)
        out = torch.empty(B, 1, n_out, (Lr + self.downsample - 1) // self.downsample, dtype=torch.float32, device=raw.device)
        dbl = C.POINTER(C.c_double)
        L.check(L.load().bx_eeg_stack_iir(_p(raw), _p(idx), _p(out), B, Lr, Craw, n_out, self.b.ctypes.data_as(dbl),
                                          self.a.ctypes.data_as(dbl), self.order, self.downsample, self.clip, self.scale, _stream()),
                "bx_eeg_stack_iir")
        return out


_DEFAULT = None


def stack_eeg(raw: torch.Tensor) -> torch.Tensor:
    """Reference defaults (19 channels already in canonical order)."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = EEGStacker()
    return _DEFAULT(raw)
