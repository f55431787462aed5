"""GPU EEG stackers.

``EEGStacker`` / ``stack_eeg``: raw windows [B, L, C_raw] -> EEGNet input [B, 1, 19, L/5] (the benchmark path).
``EEGMontageStacker`` / ``stack_eeg_montage``: raw frames [B, L, 20] -> [B, 1, 37, 3000], the notebook's native
multimodal pipeline (band-pass, NaN fill, bipolar montage, denoise, z-score, pad).

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
from .ops import _p, _stream, workspace

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


# reference XAI_Multimodality.py:115-120 (column order of the parquet frames) and :211-218 (bipolar pairs)
EEG_COLUMNS = ["Fp1", "F3", "C3", "P3", "F7", "T3", "T5", "O1", "Fz", "Cz", "Pz", "Fp2", "F4", "C4", "P4", "F8", "T4", "T6", "O2", "EKG"]
MAP_FEATURES = [("Fp1", "F7"), ("F7", "T3"), ("T3", "T5"), ("T5", "O1"), ("Fp1", "F3"), ("F3", "C3"), ("C3", "P3"), ("P3", "O1"),
                ("Fp2", "F8"), ("F8", "T4"), ("T4", "T6"), ("T6", "O2"), ("Fp2", "F4"), ("F4", "C4"), ("C4", "P4"), ("P4", "O2"),
                ("Fz", "Cz"), ("Cz", "Pz")]


class EEGMontageStacker:
    """``CombinedDataset.process_eeg`` (reference XAI_Multimodality.py:1148-1164) for a whole batch on the GPU.

    raw: CUDA float tensor [B, L, 20], columns in ``EEG_COLUMNS`` order (``eeg.values`` of each frame).
    Returns [B, 1, 37, fixed_length] float32.  ``reference_row_selection=True`` keeps the reference's row choice
    (:1126-1127, :1271-1276: the bipolar rows are stacked after all 20 raw rows but selected from index 19, so the
    output is 19 EEG channels + EKG + the first 17 differences); ``False`` gives the 19 channels + all 18 differences.
    ``mirror=True`` applies ``mirror_eeg`` (:1278-1282, cfg.AUGMENT): left and right electrode columns swapped.
    ``last_status`` (device int32) is 1 when a raw row was NaN from its first sample (the reference drops such rows).
    """

    def __init__(self, low=0.5, high=20.0, sampling_rate=200.0, fixed_length=3000, eps=1e-6, reference_row_selection=True,
                 mirror=False):
        from scipy.signal import butter
        nyq = 0.5 * sampling_rate
        self.b1, self.a1 = (np.ascontiguousarray(v, dtype=np.float64) for v in butter(5, [low / nyq, high / nyq], btype="band"))
        self.b2, self.a2 = (np.ascontiguousarray(v, dtype=np.float64) for v in butter(6, [low / nyq, high / nyq], btype="band"))
        self.fixed_length, self.eps = int(fixed_length), float(eps)
        idx = {n: i for i, n in enumerate(EEG_COLUMNS)}
        col = list(range(20))
        if mirror:
            left = ["Fp1", "F3", "C3", "P3", "F7", "T3", "T5", "O1"]
            right = ["Fp2", "F4", "C4", "P4", "F8", "T4", "T6", "O2"]
            for a, b in zip(left, right):
                col[idx[a]], col[idx[b]] = idx[b], idx[a]
        rows = [(col[i], -1) for i in range(19)]
        pairs = [(col[idx[a]], col[idx[b]]) for a, b in MAP_FEATURES]
        rows += ([(col[19], -1)] + pairs[:17]) if reference_row_selection else pairs
        self.rows = rows
        self._dev = {}
        self.last_status = None

    def __call__(self, raw: torch.Tensor) -> torch.Tensor:
        if not raw.is_cuda:
            raise RuntimeError("brainxai.EEGMontageStacker: raw EEG must be a CUDA tensor; there is no CPU path")
        if raw.dim() != 3 or raw.shape[2] != 20:
            raise RuntimeError(f"expected raw EEG frames [B, L, 20], got {tuple(raw.shape)}")
        raw = raw.contiguous().float()
        B, Lr, Craw = raw.shape
        R = len(self.rows)
        if B == 0:
            return torch.empty(0, 1, R, self.fixed_length, device=raw.device)
        key = raw.device.index
        if key not in self._dev:
            self._dev[key] = (torch.tensor([r[0] for r in self.rows], dtype=torch.int32, device=raw.device),
                              torch.tensor([r[1] for r in self.rows], dtype=torch.int32, device=raw.device))
        ra, rb = self._dev[key]
        lib = L.load()
        out = torch.empty(B, 1, R, self.fixed_length, dtype=torch.float32, device=raw.device)
        status = torch.empty(1, dtype=torch.int32, device=raw.device)
        need = lib.bx_eeg_montage_workspace(B, Lr, Craw, R)
        ws = workspace(need, raw.device)
        dbl = C.POINTER(C.c_double)
        L.check(lib.bx_eeg_montage_stack(_p(raw), _p(ra), _p(rb), _p(out), B, Lr, Craw, R, self.fixed_length,
                                         self.b1.ctypes.data_as(dbl), self.a1.ctypes.data_as(dbl), len(self.b1) - 1,
                                         self.b2.ctypes.data_as(dbl), self.a2.ctypes.data_as(dbl), len(self.b2) - 1,
                                         self.eps, _p(status), _p(ws), ws.numel(), _stream()), "bx_eeg_montage_stack")
        self.last_status = status
        return out


_MONTAGE = None


def stack_eeg_montage(raw: torch.Tensor) -> torch.Tensor:
    """Reference defaults: [B, L, 20] raw frames -> [B, 1, 37, 3000]."""
    global _MONTAGE
    if _MONTAGE is None:
        _MONTAGE = EEGMontageStacker()
    return _MONTAGE(raw)


class SpectrogramPreprocessor:
    """``CombinedDataset.process_spectrogram`` (reference XAI_Multimodality.py:1166-1204) for a whole batch on the GPU.

    raw: CUDA float tensor [B, Trows, C] -- each frame's ``to_numpy()`` without the time column (NaNs allowed).
    offsets: optional int tensor/list [B] (``spectrogram_label_offset_seconds``; the reference takes 300 COLUMNS from
    ``offset // 2``).  Returns [B, 3, 400, 300] float32 with three identical channels.  ``last_status`` (device int32):
    bit 0 = a row was entirely NaN (the reference drops it and resamples with scikit-image: not reproduced here).
    """

    def __init__(self, image_size=(400, 300), window=300, notch_freq=60.0, fs=200.0, quality=30.0, sigma=1.0, eps=1e-6):
        from scipy.signal import iirnotch, lfilter_zi
        b, a = iirnotch(notch_freq, quality, fs)
        self.b, self.a = np.ascontiguousarray(b, dtype=np.float64), np.ascontiguousarray(a, dtype=np.float64)
        self.zi = np.ascontiguousarray(lfilter_zi(b, a), dtype=np.float64)
        radius = int(4.0 * float(sigma) + 0.5)
        if radius != 4:
            raise ValueError("the kernel is built for the 9-tap (sigma = 1) gaussian of the reference")
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        self.gw = np.ascontiguousarray(phi / phi.sum(), dtype=np.float64)
        self.rows, self.cols, self.window, self.eps = int(image_size[0]), int(image_size[1]), int(window), float(eps)
        self.last_status = None

    def __call__(self, raw: torch.Tensor, offsets=None) -> torch.Tensor:
        if not raw.is_cuda:
            raise RuntimeError("brainxai.SpectrogramPreprocessor: raw spectrograms must be a CUDA tensor; there is no CPU path")
        if raw.dim() != 3:
            raise RuntimeError(f"expected raw spectrogram frames [B, Trows, C], got {tuple(raw.shape)}")
        raw = raw.contiguous().float()
        B, Trows, Cc = raw.shape
        if B == 0:
            return torch.empty(0, 3, self.rows, self.cols, device=raw.device)
        off = None
        if offsets is not None:
            off = torch.as_tensor(offsets, dtype=torch.int32).to(raw.device).contiguous()
            if off.numel() != B or int(off.min()) < 0:
                raise RuntimeError("offsets must be B non-negative integers")
        lib = L.load()
        out = torch.empty(B, 3, self.rows, self.cols, dtype=torch.float32, device=raw.device)
        status = torch.empty(1, dtype=torch.int32, device=raw.device)
        ws = workspace(lib.bx_spec_preprocess_workspace(B, self.rows, self.cols), raw.device)
        dbl = C.POINTER(C.c_double)
        L.check(lib.bx_spec_preprocess(_p(raw), _p(off), _p(out), B, Trows, Cc, self.rows, self.cols, self.window,
                                       self.b.ctypes.data_as(dbl), self.a.ctypes.data_as(dbl), self.zi.ctypes.data_as(dbl),
                                       self.gw.ctypes.data_as(dbl), self.eps, _p(status), _p(ws), ws.numel(), _stream()),
                "bx_spec_preprocess")
        self.last_status = status
        return out


_SPECPREP = None


def preprocess_spectrograms(raw: torch.Tensor, offsets=None) -> torch.Tensor:
    """Reference defaults: [B, Trows, 400] parquet values -> [B, 3, 400, 300]."""
    global _SPECPREP
    if _SPECPREP is None:
        _SPECPREP = SpectrogramPreprocessor()
    return _SPECPREP(raw, offsets)


class SpectrogramRegionStacker:
    """The benchmark's spectrogram input (BASELINE.json: [B, 4, 128, 256]) from raw parquet values, on the GPU: the four 100-bin
    regions of a Kaggle HMS spectrogram (LL, RL, RP, LP = 400 columns) become four channel planes.  The reference has no 4-plane
    pipeline (its CNN takes one 400 x 300 image tiled to 3 channels); this composes the reference's own helpers on that layout:
    a window of ``window`` time rows from ``offset // 2`` (zero padded; XAI_Multimodality.py:1178-1183), ``normalize_signal``
    (root/src/utils/data_utils.py:133-136) over the sample and ``resample_spectrogram`` (data_utils.py:145-147: scikit-image's
    anti-aliased bilinear resize, restated from scikit-image 0.24: gaussian pre-filter with sigma = (s - 1) / 2 where an axis
    shrinks, then an order-1 zoom) of every region to ``out_hw``.

    raw: CUDA float tensor [B, Trows, regions * bins] (NaNs allowed); offsets: optional int tensor / list [B].
    Returns float32 [B, regions, out_h, out_w]."""

    def __init__(self, out_hw=(128, 256), window=300, regions=4, eps=1e-6):
        self.out_hw, self.window, self.regions, self.eps = (int(out_hw[0]), int(out_hw[1])), int(window), int(regions), float(eps)
        self._g = {}

    @staticmethod
    def _gauss(n_in, n_out, truncate=4.0):
        sigma = max(0.0, (n_in / n_out - 1.0) / 2.0)
        radius = int(truncate * sigma + 0.5)
        if radius == 0:
            return np.ones(1, dtype=np.float64), 0
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        return np.ascontiguousarray(phi / phi.sum(), dtype=np.float64), radius

    def __call__(self, raw: torch.Tensor, offsets=None) -> torch.Tensor:
        if not raw.is_cuda:
            raise RuntimeError("brainxai.SpectrogramRegionStacker: raw spectrograms must be a CUDA tensor; there is no CPU path")
        if raw.dim() != 3 or raw.shape[2] % self.regions:
            raise RuntimeError(f"expected raw spectrogram frames [B, Trows, {self.regions} * bins], got {tuple(raw.shape)}")
        raw = raw.contiguous().float()
        B, Trows, Cc = raw.shape
        Ho, Wo = self.out_hw
        if B == 0:
            return torch.empty(0, self.regions, Ho, Wo, device=raw.device)
        off = None
        if offsets is not None:
            off = torch.as_tensor(offsets, dtype=torch.int32).to(raw.device).contiguous()
            if off.numel() != B or int(off.min()) < 0:
                raise RuntimeError("offsets must be B non-negative integers")
        key = (Cc // self.regions, self.window)
        if key not in self._g:
            self._g[key] = (self._gauss(key[0], Ho), self._gauss(key[1], Wo))
        (gy, ry), (gx, rx) = self._g[key]
        if ry > 8 or rx > 8:
            raise RuntimeError("SpectrogramRegionStacker: down-scaling factor too large for the kernel's anti-aliasing radius (8)")
        lib = L.load()
        out = torch.empty(B, self.regions, Ho, Wo, dtype=torch.float32, device=raw.device)
        ws = workspace(lib.bx_spec_regions_workspace(B), raw.device)
        dbl = C.POINTER(C.c_double)
        L.check(lib.bx_spec_regions(_p(raw), _p(off), _p(out), B, Trows, Cc, self.regions, self.window, Ho, Wo, gy.ctypes.data_as(dbl), ry,
                                    gx.ctypes.data_as(dbl), rx, self.eps, _p(ws), ws.numel(), _stream()), "bx_spec_regions")
        return out


_REGIONS = None


def stack_spectrogram_regions(raw: torch.Tensor, offsets=None) -> torch.Tensor:
    """Benchmark defaults: [B, Trows, 400] parquet values -> [B, 4, 128, 256]."""
    global _REGIONS
    if _REGIONS is None:
        _REGIONS = SpectrogramRegionStacker()
    return _REGIONS(raw, offsets)


# ------------------------------------------------------------------------------------------------
class StagingRing:
    """SURVEY 8(f) rank 2, the staging half: decoded parquet values and raw EEG windows travel host -> GPU through a ring of pinned
    host slots, the H2D copies on a COPY stream, the GPU stackers on a PREP stream, while the training step of an earlier batch
    runs on the caller's stream -- the reference does this work in ``num_workers`` DataLoader processes and a synchronous
    ``.to(device)`` per batch (XAI_Multimodality.py:1132-1146, 1900; dataset.py:182-228).

        ring = StagingRing({"eeg": (64, 10000, 19), "spec": (64, 320, 400)}, transform=lambda d: (stack_eeg(d["eeg"]), stack_spectrogram_regions(d["spec"])))
        slot = ring.acquire()                  # a free slot: dict of PINNED host tensors; the loader decodes straight into them
        ...fill slot.host["eeg"], slot.host["spec"]...
        ring.submit(slot)                      # H2D on the copy stream, then ``transform`` on the prep stream; returns at once
        batch = ring.pop()                     # oldest submitted slot: the current stream waits for its outputs; batch.outputs = transform's result
        ...train on batch.outputs...
        ring.release(batch)                    # the slot's device buffers may be overwritten once the current stream gets here

    A slot cycles FREE -> FILLING (acquire) -> IN_FLIGHT (submit) -> READY (pop) -> FREE (release).  Ordering is carried by events:
    ``h2d`` (copy stream: the pinned buffers may be refilled), ``ready`` (prep stream: outputs complete), ``consumed`` (caller's
    stream at release: the copy stream waits for it before overwriting the slot's device buffers).  ``acquire`` blocks the HOST only
    on the h2d event of the slot it hands out.  ``device=None`` / a CPU device runs everything synchronously (slot logic for tests,
    no GPU needed); the arithmetic itself has no CPU path.

    ``threaded=True`` (default on a GPU): ``submit`` hands the slot to a feeder thread that issues the copies and the transform's
    launches.  Measured on MI355X / ROCm 7.2: a pinned ``hipMemcpyAsync`` of tens of MB returns only when the copies queued before
    it on that stream have drained, so the issuing thread is paced by PCIe (1.44 ms per 81 MB batch) -- issued from the training
    thread that time ADDS to its Python work per step (1.9 ms per iteration against a 1.54 ms step); from a feeder thread (the GIL
    is released inside the call) the two overlap."""

    FREE, FILLING, IN_FLIGHT, READY = range(4)

    class Slot:
        def __init__(self, index):
            self.index, self.state = index, StagingRing.FREE
            self.host, self.dev, self.outputs = {}, {}, None
            self.h2d = self.ready = self.consumed = None
            self.issued = None                              # threading.Event: the feeder thread has queued this slot's GPU work

    def __init__(self, shapes, transform=None, slots=3, device="cuda", dtype=torch.float32, threaded=None):
        if slots < 2:
            raise ValueError("StagingRing needs at least two slots (one filling while one is in flight)")
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.transform = transform
        self.order = []                                     # submitted, not yet popped (FIFO)
        self._next = 0                                      # acquire() walks the slots round-robin: the slot handed out is the one
                                                            # released LONGEST ago, so its `consumed` event is long past when its H2D starts
        self.log = []                                       # (event, slot index) history, for tests and debugging
        self.slots = [StagingRing.Slot(i) for i in range(slots)]
        for sl in self.slots:
            for name, shape in shapes.items():
                host = torch.empty(tuple(shape), dtype=dtype)
                sl.host[name] = host.pin_memory() if self.cuda else host
                sl.dev[name] = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self._thread = self._queue = self._error = None
        if self.cuda:
            self.copy_stream, self.prep_stream = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)
            if threaded is None or threaded:
                import queue
                import threading
                self._queue = queue.Queue()
                self._thread = threading.Thread(target=self._feeder, daemon=True)
                self._thread.start()

    def _feeder(self):
        torch.cuda.set_device(self.device)
        while True:
            item = self._queue.get()
            if item is None:
                return
            slot, consumer = item
            try:
                self._issue(slot, consumer)
            except Exception as exc:                        # noqa: BLE001  (re-raised by pop())
                self._error = exc
            finally:
                slot.issued.set()

    def close(self):
        """Stop the feeder thread (outstanding submissions are issued first)."""
        if self._thread is not None:
            self._queue.put(None)
            self._thread.join()
            self._thread = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # noqa: BLE001  (interpreter shutdown)
            pass

    def _issue(self, slot, consumed):
        with torch.cuda.stream(self.copy_stream):
            if consumed is not None:
                self.copy_stream.wait_event(consumed)       # the last consumer of this slot's device buffers is done
            for name, host in slot.host.items():
                slot.dev[name].copy_(host, non_blocking=True)
            h2d = torch.cuda.Event()
            h2d.record(self.copy_stream)
        with torch.cuda.stream(self.prep_stream):
            self.prep_stream.wait_event(h2d)
            slot.outputs = self.transform(slot.dev) if self.transform is not None else tuple(slot.dev.values())
            ready = torch.cuda.Event()
            ready.record(self.prep_stream)
        slot.h2d, slot.ready = h2d, ready

    def acquire(self):
        """A FREE slot in FILLING state (round-robin), or None when every slot is filling, in flight or ready."""
        n = len(self.slots)
        for k in range(n):
            sl = self.slots[(self._next + k) % n]
            if sl.state == StagingRing.FREE:
                if sl.issued is not None:
                    sl.issued.wait()
                if sl.h2d is not None:
                    sl.h2d.synchronize()                    # the previous copy out of the pinned buffers has finished
                sl.state = StagingRing.FILLING
                self._next = (sl.index + 1) % n
                self.log.append(("acquire", sl.index))
                return sl
        return None

    def submit(self, slot):
        if slot.state != StagingRing.FILLING:
            raise RuntimeError(f"StagingRing.submit: slot {slot.index} was not acquired (state {slot.state})")
        if self.cuda:
            if self._thread is not None:
                import threading
                slot.issued = threading.Event()
                self._queue.put((slot, slot.consumed))
            else:
                self._issue(slot, slot.consumed)
        else:
            for name, host in slot.host.items():
                slot.dev[name].copy_(host)
            slot.outputs = self.transform(slot.dev) if self.transform is not None else tuple(slot.dev.values())
        slot.state = StagingRing.IN_FLIGHT
        self.order.append(slot)
        self.log.append(("submit", slot.index))

    def pop(self):
        """The oldest submitted slot; the CURRENT stream waits for its outputs (no host synchronisation)."""
        if not self.order:
            raise RuntimeError("StagingRing.pop: nothing was submitted")
        slot = self.order.pop(0)
        if self.cuda:
            if slot.issued is not None:
                slot.issued.wait()                          # the feeder thread has queued the slot's copies and launches
            if self._error is not None:
                err, self._error = self._error, None
                raise RuntimeError(f"StagingRing: the feeder thread failed: {err}") from err
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot.ready)
            for t in (slot.outputs if isinstance(slot.outputs, (tuple, list)) else (slot.outputs,)):
                if torch.is_tensor(t):
                    t.record_stream(cur)                    # allocated on the prep stream, read on this one
        slot.state = StagingRing.READY
        self.log.append(("pop", slot.index))
        return slot

    def release(self, slot):
        if slot.state != StagingRing.READY:
            raise RuntimeError(f"StagingRing.release: slot {slot.index} is not checked out (state {slot.state})")
        if self.cuda:
            slot.consumed = torch.cuda.Event()
            slot.consumed.record(torch.cuda.current_stream(self.device))
        slot.outputs = None
        slot.state = StagingRing.FREE
        self.log.append(("release", slot.index))

    def in_flight(self):
        return len(self.order)
