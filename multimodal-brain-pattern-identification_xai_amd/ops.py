"""torch.autograd glue over the C ABI: one Function per fused stage of the hot path.

PyTorch is used here for device memory (caching allocator), the current HIP stream and autograd
bookkeeping only; every arithmetic step is a launch into libbrainxai.so.  Activations of the 2-D CNN
are channels-last tensors [B, H, W, C] in the compute dtype (float32 or bfloat16); parameters and
their gradients are float32.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib as L

CONV_ALGO = L.BX_ALGO_AUTO        # module-level switch used by tests to force the direct / MFMA kernels
WGRAD_ALGO = L.BX_ALGO_AUTO
CONV_PROFILE = None               # bench.py sets a list: every conv launch appends (kind, start_event, end_event, meta); meta =
                                  # (form, B, H, W, Ci, Co) with form "conv" | "pair" (two layers, Co = both widths) | "conv3+pool"
# Multi-stream overlap (most kernels of this model are latency-bound and leave CUs idle): weight-gradient kernels run
# on a side stream beside the data-gradient chain, the EEG branch beside the spectrogram branch.  join_side_streams()
# must run before anything consumes the gradients (FlatAdamW.step / DataParallel.sync_gradients do it).
# Measured on MI355X (round 1): +2 % under hipGraph replay, -10 % when launching eagerly (host-side stream switches),
# so it is OFF by default; BX_OVERLAP=1 turns it on.
import os as _os
OVERLAP = _os.environ.get("BX_OVERLAP", "0") == "1"
# The EEG branch on a side stream beside the spectrogram branch (one fork / join per direction; round 3).  BX_OVERLAP_EEG: 0 = never,
# 1 (default) = inside captured hipGraphs (GraphedTrainStep, GradCamSweep: two parallel chains of the graph), 2 = also when launching
# eagerly (host-side stream switches make that slower than the serial order).  Same kernels, same arithmetic: the overlapped step
# walks the serial trajectory bit for bit (tests/test_gpu_bench_config.py, tools/trajectory_check.py).
OVERLAP_EEG = int(_os.environ.get("BX_OVERLAP_EEG", "1"))
# ... and, opt-in (BX_OVERLAP_EEG_DDP=1), inside the one-graph data-parallel capture, where RCCL's stream is a third branch: 1.50 -> 1.41 ms
# on a 1-rank group, but destroy_process_group() then aborted in c10d at teardown in 2 of 4 full-suite runs (never without it)
OVERLAP_EEG_DDP = _os.environ.get("BX_OVERLAP_EEG_DDP", "0") == "1"


def overlap_eeg_now() -> bool:
    if CONV_PROFILE is not None or not (OVERLAP_EEG >= 2 or (OVERLAP_EEG == 1 and torch.cuda.is_current_stream_capturing())):
        return False
    # Not in a process that holds a process group (unless BX_OVERLAP_EEG_DDP=1): with multi-branch graphs captured while RCCL's communicator
    # exists, destroy_process_group() aborted in c10d at teardown in 3 of 5 full-suite runs -- with any of the data-parallel step forms,
    # not only the one-graph capture -- and never in the ~15 runs before the side stream existed.
    import torch.distributed as dist
    return OVERLAP_EEG_DDP or not (dist.is_available() and dist.is_initialized())


def fork_eeg(device, *read_on_side):
    """(main, side) streams with the side stream waiting for the main one; tensors the side stream will read are recorded on it."""
    cur = torch.cuda.current_stream()
    side = side_stream("eeg", device)
    side.wait_stream(cur)
    for t in read_on_side:
        if t is not None:
            t.record_stream(side)
    return cur, side
FUSED_HEAD = _os.environ.get("BX_FUSED_HEAD", "1") == "1"    # MultimodalModel: GAP+fc, dense and the fusion head in one launch
_SIDE = {}


def side_stream(name: str, device) -> "torch.cuda.Stream":
    key = (name, device.index if device.index is not None else torch.cuda.current_device())
    st = _SIDE.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[key] = st
    return st


def join_side_streams(device=None):
    cur = torch.cuda.current_stream()
    for (name, idx), st in _SIDE.items():
        if device is None or idx == (device.index if device.index is not None else torch.cuda.current_device()):
            cur.wait_stream(st)


_JOIN_QUEUED = [False]


def _join_after_backward():
    """Queue ONE engine callback per backward pass: when autograd finishes, the caller's stream waits for the side
    streams, so `p.grad` is safe to read right after `loss.backward()` returns (eager users, torch optimizers)."""
    if _JOIN_QUEUED[0]:
        return

    def _cb():
        _JOIN_QUEUED[0] = False
        join_side_streams()

    _JOIN_QUEUED[0] = True
    torch.autograd.Variable._execution_engine.queue_callback(_cb)


class _Timed:
    """HIP-event bracket around one launch on the current stream (only active while CONV_PROFILE is a list)."""

    def __init__(self, kind, meta=None):
        self.kind, self.meta = kind, meta

    def __enter__(self):
        if CONV_PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if CONV_PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            CONV_PROFILE.append((self.kind, self.e0, e1, self.meta))


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"brainxai: {what} must live on the GPU (got {t.device}); this package has no CPU path")


def bx_dtype(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return L.BX_F32
    if dt == torch.bfloat16:
        return L.BX_BF16
    raise RuntimeError(f"brainxai: unsupported compute dtype {dt}")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_WS = {}


_WS_RETIRED = []


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): all uses of one buffer are ordered on its stream."""
    if OVERLAP_EEG == 1 and not torch.cuda.is_current_stream_capturing() and torch.cuda.current_stream().cuda_stream != side_stream("eeg", device).cuda_stream:
        # the EEG branch's side stream gets its scratch buffer NOW, outside any capture: allocated for the first time inside a captured
        # step it would live in that graph's private pool for the rest of the process
        sk = (device.index if device.index is not None else torch.cuda.current_device(), side_stream("eeg", device).cuda_stream)
        sb = _WS.get(sk)
        if sb is None or sb.numel() < nbytes:
            if sb is not None:
                _WS_RETIRED.append(sb)
            with torch.cuda.stream(side_stream("eeg", device)):
                _WS[sk] = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _WS_RETIRED.append(buf)       # a captured hipGraph may still point at the smaller buffer: keep it mapped
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


_SEED = {}


def seed_state(device, lane: str = "spec") -> torch.Tensor:
    """Dropout counter of a (device, lane); the EEG branch has its own lane because it runs on its own stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), lane)
    st = _SEED.get(key)
    if st is None:
        base = (torch.initial_seed() & 0x7FFFFFFFFFFF) + (0 if lane == "spec" else 0x5EED0000)
        st = torch.full((1,), base, dtype=torch.int64, device=device)
        _SEED[key] = st
    return st


def manual_seed(seed: int, device=None):
    """Reset the dropout counter streams of ``device`` (default: current device)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    seed_state(device, "spec").fill_(int(seed))
    seed_state(device, "eeg").fill_(int(seed) + 0x5EED0000)


def next_seed(device, lane: str = "spec") -> torch.Tensor:
    out = torch.empty(1, dtype=torch.int64, device=device)
    L.check(L.load().bx_seed_next(_p(seed_state(device, lane)), _p(out), _stream()), "bx_seed_next")
    return out


def next_seed_pair(device):
    """(spectrogram-lane seed, EEG-lane seed) from ONE launch; same values as two next_seed() calls."""
    out = torch.empty(2, dtype=torch.int64, device=device)
    L.check(L.load().bx_seed_next2(_p(seed_state(device, "spec")), out.data_ptr(), _p(seed_state(device, "eeg")), out.data_ptr() + 8, _stream()),
            "bx_seed_next2")
    return out[0:1], out[1:2]


# Gradient arena hook: the trainer (FlatAdamW) registers, per parameter, the slice of its flat fp32 gradient buffer, so
# weight-gradient kernels write there directly.  Entries are keyed by the parameter's storage address (saved tensors are new
# objects, the storage identifies the parameter) and carry a weak reference to the Parameter: a dead reference (the model was
# dropped and the address reused) is ignored, and a live one lets new_grad() refuse to ALIAS an existing ``.grad``.
import weakref as _weakref

_GRAD_VIEW = {}
_GRAD_KEYS = {}
_GRAD_NEXT = [0]


def register_grad_views(params, flat: torch.Tensor):
    """Returns a registration key for unregister_grad_views()."""
    off, ptrs = 0, []
    for p_ in params:
        n = p_.numel()
        _GRAD_VIEW[p_.data_ptr()] = (flat, off, tuple(p_.shape), _weakref.ref(p_))
        ptrs.append(p_.data_ptr())
        off += n
    _GRAD_NEXT[0] += 1
    _GRAD_KEYS[_GRAD_NEXT[0]] = (ptrs, flat)
    return _GRAD_NEXT[0]


def unregister_grad_views(key):
    ent = _GRAD_KEYS.pop(key, None) if key is not None else None
    if ent is None:
        return
    ptrs, flat = ent
    for ptr in ptrs:
        cur = _GRAD_VIEW.get(ptr)
        if cur is not None and cur[0] is flat:
            del _GRAD_VIEW[ptr]


def clear_grad_views():
    _GRAD_VIEW.clear()
    _GRAD_KEYS.clear()


def new_grad(param: torch.Tensor) -> torch.Tensor:
    ent = _GRAD_VIEW.get(param.data_ptr())
    if ent is None:
        return torch.empty_like(param, dtype=torch.float32)
    flat, off, shape, ref = ent
    owner = ref()
    if owner is None or owner.data_ptr() != param.data_ptr() or tuple(owner.shape) != tuple(param.shape):
        del _GRAD_VIEW[param.data_ptr()]                  # stale: that optimizer's model is gone, the address was reused
        return torch.empty_like(param, dtype=torch.float32)
    view = flat.narrow(0, off, param.numel()).view(shape)
    if owner.grad is not None and owner.grad.data_ptr() == view.data_ptr():
        # the kernels WRITE the arena slice and autograd would then add the slice to itself (p.grad += g, same memory):
        # a second backward without zero_grad(), zero_grad(set_to_none=False), or a parameter used twice in one graph
        raise RuntimeError("brainxai: this parameter's .grad already is its gradient-arena slice; gradient accumulation over several "
                           "backward passes is not supported with FlatAdamW -- call optimizer.zero_grad() (set_to_none=True) first")
    return view


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------
def to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 NCHW [B,C,H,W] (any strides) -> compute-dtype NHWC [B,H,W,pad8(C)] with zero padding."""
    _require_gpu(x, "input")
    B, Cc, H, W = x.shape
    src = x.detach().to(torch.float32).contiguous()
    out = torch.empty(B, H, W, pad8(Cc), dtype=dtype, device=x.device)
    L.check(L.load().bx_nchw_to_nhwc(_p(src), _p(out), B, Cc, H, W, pad8(Cc), bx_dtype(dtype), _stream()), "bx_nchw_to_nhwc")
    return out


def to_nchw_f32(x_nhwc: torch.Tensor, channels: int) -> torch.Tensor:
    B, H, W, Cs = x_nhwc.shape
    out = torch.empty(B, channels, H, W, dtype=torch.float32, device=x_nhwc.device)
    L.check(L.load().bx_nhwc_to_nchw(_p(x_nhwc), _p(out), B, channels, H, W, Cs, bx_dtype(x_nhwc.dtype), _stream()), "bx_nhwc_to_nchw")
    return out


class InputLayout(torch.autograd.Function):
    """Differentiable NCHW fp32 -> NHWC(pad 8) compute-dtype conversion (gradient flows back for saliency / IG)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.channels = x.shape[1]
        return to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return to_nchw_f32(g.contiguous(), ctx.channels), None


def _pack(w: torch.Tensor, flip: bool, dtype=None):
    """fp32 OIHW -> library operand(s). Returns (packed_f32|None, packed_mfma|None, I_p, O_p, layout).
    ``layout`` names what packed_mfma holds: "bf16" (bf16 activations) or "split" (fp32 activations on the matrix cores: h + m + l
    images, csrc/conv3x3_split.hip).  With a compute dtype given and an MFMA layout available only that operand is produced;
    dtype=None produces the fp32 operand and the bf16 one."""
    lib = L.load()
    co, ci = w.shape[0], w.shape[1]
    i_log, o_log = (co, ci) if flip else (ci, co)
    ip, op = pad8(i_log), pad8(o_log)
    pm, layout = None, None
    if CONV_ALGO != L.BX_ALGO_DIRECT:
        if dtype == torch.float32:
            nb, layout = lib.bx_conv3x3_packed_split_bytes(ip, op), "split"
        else:
            nb, layout = lib.bx_conv3x3_packed_mfma_bytes(ip, op), "bf16"
        if nb:
            pm = torch.empty(nb, dtype=torch.uint8, device=w.device)
        else:
            layout = None
    need_f32 = dtype is None or pm is None
    pf = torch.empty(9 * ip * op, dtype=torch.float32, device=w.device) if need_f32 else None
    if layout == "split":
        L.check(lib.bx_conv3x3_pack_split(_p(w), _p(pm), co, ci, ip, op, 1 if flip else 0, _stream()), "bx_conv3x3_pack_split")
    else:
        L.check(lib.bx_conv3x3_pack(_p(w), _p(pf), _p(pm), co, ci, ip, op, 1 if flip else 0, _stream()), "bx_conv3x3_pack")
    return pf, pm, ip, op, layout


# Bumped by everything in this library that rewrites parameters behind torch's back (FlatAdamW's kernel, replayed training graphs);
# together with the tensors' own version counters it tells a PackPlan whether its packed operands are still those of the weights.
PARAM_EPOCH = [0]


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


# Evaluation-mode forwards skip the weight-packing jobs (PackPlan.run(reuse=True)) only inside a scope that owns the contract
# "parameters are frozen here": attribution passes (explain._eval_frozen) and GradCamSweep.  Elsewhere every forward re-packs, because
# a parameter rewritten through a `.data` view (old-style code, `dist.broadcast(p.data, 0)`) changes neither its version counter nor
# PARAM_EPOCH.  Entering the scope bumps the epoch, so the first forward inside it always packs.
PACK_REUSE = [False]


class pack_reuse:
    def __enter__(self):
        self.prev = PACK_REUSE[0]
        bump_param_epoch()
        PACK_REUSE[0] = True

    def __exit__(self, *exc):
        PACK_REUSE[0] = self.prev


# data_ptr of a static graph input -> address of the DEVICE word its kernels read the batch's address from (GradCamSweep: a replay
# is pointed at the caller's batch with bx_store_u64x2 instead of copying the batch into the static buffer)
INPUT_SLOTS = {}
_SLOT_ADDR = {}


def _take_slot(t):
    """Slot address registered for tensor ``t`` (or None); the registration is marked "used" so that the sweep knows this input is
    read through the slot by every captured kernel that touches it."""
    if not INPUT_SLOTS:
        return None
    ptr = t.data_ptr()
    ent = INPUT_SLOTS.get(ptr)
    if ent is None:
        return None
    if ent != "used":
        _SLOT_ADDR[ptr] = ent
        INPUT_SLOTS[ptr] = "used"
    return _SLOT_ADDR[ptr]


class PackPlan:
    """All MFMA weight operands (forward + data-gradient) of a list of 3x3 convolutions, packed by ONE launch.

    Built once per set of parameter storages (stable under FlatAdamW); ``run()`` re-packs every step because the
    weights change -- except ``run(reuse=True)`` (evaluation mode), which skips the pack jobs while no parameter changed since
    the last executed pack (``fresh()``).  ``get(i, flip)`` returns the operand tuple _conv expects, or None when the MFMA path
    does not cover that shape (the caller then packs it individually)."""

    def __init__(self, weights, dtype=torch.bfloat16):
        import ctypes
        lib = L.load()
        dev = weights[0].device
        self.dtype = dtype
        split = dtype == torch.float32                  # fp32 storage: h + m + l images (csrc/conv3x3_split.hip), job bit 1
        nbytes = lib.bx_conv3x3_packed_split_bytes if split else lib.bx_conv3x3_packed_mfma_bytes
        self.key = tuple(w.data_ptr() for w in weights) + (dtype,)
        self.weights, self.packed_state = list(weights), None
        entries, off = [], 0
        for i, w in enumerate(weights):
            co, ci = w.shape[0], w.shape[1]
            for flip in (False, True):
                i_log, o_log = (co, ci) if flip else (ci, co)
                ip, op = pad8(i_log), pad8(o_log)
                nb = nbytes(ip, op)
                if nb:
                    entries.append((i, flip, w, co, ci, ip, op, off, nb))
                    off += (nb + 255) // 256 * 256
        self.buf = torch.empty(max(off, 256), dtype=torch.uint8, device=dev)
        self.views, jobs, blk = {}, (L.PackJob * len(entries))(), 0
        for j, (i, flip, w, co, ci, ip, op, o, nb) in enumerate(entries):
            view = self.buf.narrow(0, o, nb)
            self.views[(i, flip)] = (None, view, ip, op, "split" if split else "bf16")
            jobs[j] = L.PackJob(w.data_ptr(), view.data_ptr(), co, ci, ip, op, (1 if flip else 0) | (2 if split else 0), blk)
            blk += max(1, (nb // (6 if split else 2) + 2047) // 2048)   # 2048 packed elements (8 per thread) per workgroup
        self.njobs, self.nblocks = len(entries), blk
        raw = bytes(memoryview(jobs)) if entries else b"\0" * 8
        self.jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def _state(self):
        return (PARAM_EPOCH[0],) + tuple(w._version for w in self.weights)

    def fresh(self):
        """True while the packed operands are those of the current weights (nothing rewrote a parameter since the last pack that
        actually executed; a pack that was only captured into a graph does not count)."""
        return self.packed_state is not None and self.packed_state == self._state()

    def _mark(self):
        self.packed_state = None if torch.cuda.is_current_stream_capturing() else self._state()

    def run_step(self, x_nchw=None):
        """run() + next_seed_pair() in ONE launch (the launch that opens a multimodal training step): returns
        (batch in the internal layout or None, spectrogram-lane seed, EEG-lane seed)."""
        dev = self.buf.device
        seeds = torch.empty(2, dtype=torch.int64, device=dev)
        out, src, dims = None, None, (0, 0, 0, 0, 8)
        if x_nchw is not None:
            assert self.dtype == torch.bfloat16, "the packing launch converts the batch to the bf16 layout only"
            B, Cc, H, W = x_nchw.shape
            src = x_nchw.detach().to(torch.float32).contiguous()
            out = torch.empty(B, H, W, pad8(Cc), dtype=torch.bfloat16, device=dev)
            dims = (B, Cc, H, W, pad8(Cc))
        L.check(L.load().bx_conv3x3_pack_many_step(_p(self.jobs_dev), self.njobs, self.nblocks, _p(src), _p(out), *dims,
                                                   _p(seed_state(dev, "spec")), seeds.data_ptr(), _p(seed_state(dev, "eeg")),
                                                   seeds.data_ptr() + 8, _stream()), "bx_conv3x3_pack_many_step")
        self._mark()
        return (None if out is None else out.permute(0, 3, 1, 2)), seeds[0:1], seeds[1:2]

    def run(self, x_nchw=None, reuse=False):
        """Pack every operand; with ``x_nchw`` (fp32 NCHW, no gradient needed) the same launch also produces the batch in the
        internal bf16 channels-last layout and returns it as a logical-NCHW view.  ``reuse=True``: skip the pack jobs while
        ``fresh()`` (the launch is then the layout conversion alone, or nothing)."""
        skip = reuse and self.fresh()
        if x_nchw is not None and self.njobs:
            assert self.dtype == torch.bfloat16, "the packing launch converts the batch to the bf16 layout only"
            B, Cc, H, W = x_nchw.shape
            src = x_nchw.detach().to(torch.float32).contiguous()
            out = torch.empty(B, H, W, pad8(Cc), dtype=torch.bfloat16, device=x_nchw.device)
            slot = _take_slot(src)
            if skip or slot:
                L.check(L.load().bx_conv3x3_pack_layout_ex(None if skip else _p(self.jobs_dev), 0 if skip else self.njobs,
                                                           0 if skip else self.nblocks, _p(src), slot, _p(out), B, Cc, H, W, pad8(Cc),
                                                           _stream()), "bx_conv3x3_pack_layout_ex")
            else:
                L.check(L.load().bx_conv3x3_pack_many_layout(_p(self.jobs_dev), self.njobs, self.nblocks, _p(src), _p(out), B, Cc, H, W,
                                                             pad8(Cc), _stream()), "bx_conv3x3_pack_many_layout")
            if not skip:
                self._mark()
            return out.permute(0, 3, 1, 2)
        if self.njobs and not skip:
            L.check(L.load().bx_conv3x3_pack_many(_p(self.jobs_dev), self.njobs, self.nblocks, _stream()), "bx_conv3x3_pack_many")
            self._mark()
        return None

    def get(self, i, flip):
        return self.views.get((i, flip))


def _conv(x, packed, bias, mask_src, addend, relu: bool, dtype, carry: bool = False, mask_bits: bool = False):
    """carry=True: a pending chained weight-gradient sum of this device rides in the convolution's launch (bx_conv3x3_carry).
    mask_bits=True: ``mask_src`` is the bit form of the ReLU decisions written by bx_conv3x3_pair (BX_EPI_MASK_BITS)."""
    pf, pm, ip, op, layout = packed
    B, H, W, Ci = x.shape
    if Ci != ip:
        raise RuntimeError(f"brainxai: conv input has {Ci} channels, packed weights expect {ip}")
    y = torch.empty(B, H, W, op, dtype=dtype, device=x.device)
    if pm is not None and layout != ("bf16" if x.dtype == torch.bfloat16 else "split"):
        pm = None                                           # an MFMA operand packed for the other storage type: direct kernel
        if pf is None:
            raise RuntimeError(f"brainxai: weights were packed for {layout} activations, the input is {x.dtype}")
    algo = CONV_ALGO if pm is not None else L.BX_ALGO_DIRECT
    st = _wg_chain_state(x.device) if carry and WGRAD_CARRY else None
    flags = (L.BX_EPI_RELU if relu else 0) | (L.BX_EPI_MASK_BITS if mask_bits else 0)
    with _Timed("fwd" if bias is not None else "dgrad", ("conv", B, H, W, Ci, op)):
        if st is not None and st.pend.valid and st.stream == _stream():
            L.check(L.load().bx_conv3x3_carry(_p(x), _p(pf), _p(pm), _p(bias), _p(mask_src), _p(addend), _p(y), B, H, W, Ci, op,
                                              bx_dtype(dtype), flags, algo, C.byref(st.pend), _stream()), "bx_conv3x3_carry")
            st.keep = None
        else:
            L.check(L.load().bx_conv3x3(_p(x), _p(pf), _p(pm), _p(bias), _p(mask_src), _p(addend), _p(y), B, H, W, Ci, op,
                                        bx_dtype(dtype), flags, algo, _stream()), "bx_conv3x3")
    return y


# Chained weight gradients: inside one Block's backward the sum of a layer's partials rides in the next layer's launch
# (bx_conv3x3_wgrad_chained); BlockFn.backward finishes the chain before it returns (autograd may copy or accumulate a
# returned gradient right away, so nothing may be pending then).  Per device: the pending descriptor, two partial buffers
# used alternately (the pending partials must outlive the next launch) and the tensors to keep alive meanwhile.
WGRAD_CHAIN = _os.environ.get("BX_WGRAD_CHAIN", "1") == "1"
# the last pending sum of a Block's backward rides in conv1's data-gradient launch; 0 = a k_wgrad_reduce3 launch per Block
WGRAD_CARRY = _os.environ.get("BX_WGRAD_CARRY", "1") == "1"
# batch-statistics finalizes inside the kernels that produce the partial sums (bxTailDesc.sync); 0 = separate finalize launches
TAIL_IN_LAUNCH = _os.environ.get("BX_TAIL_IN_LAUNCH", "1") == "1"
# conv3 of a Block pools and sums the batch statistics in its epilogue (bx_block_conv3_tail_fwd); 0 = conv3, then the pooling kernel
FUSE_POOL = _os.environ.get("BX_FUSE_POOL", "1") == "1"
# ... and writes where each pooled element's gradient goes as a nibble (bxTailDesc.route) instead of storing conv3's output; 0 = store y3
TAIL_ROUTE = _os.environ.get("BX_TAIL_ROUTE", "1") == "1"
# the pair launch also writes the ReLU decisions as bits and the data gradients read those (BX_EPI_MASK_BITS); 0 = read the activations
MASK_BITS = _os.environ.get("BX_MASK_BITS", "1") == "1"
# stage 1's conv1 and conv2 in one launch (bx_conv3x3_pair); 0 = two bx_conv3x3 launches
CONV_PAIR = _os.environ.get("BX_CONV_PAIR", "1") == "1"
# EEGNet front end without the conv1 output tensor (bxEegDesc.collapse); 0 = conv1, BatchNorm1 and the electrode mix layer by layer
EEG_COLLAPSE = _os.environ.get("BX_EEG_COLLAPSE", "1") == "1"
_WG_CHAIN = {}


def _wg_chain_state(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _WG_CHAIN.get(key)
    if st is None:
        st = SimpleNamespace(pend=L.WgradPending(), ring=[None, None], slot=0, keep=None, stream=None, retired=[])
        _WG_CHAIN[key] = st
    return st


def wgrad_flush(device=None):
    """Sum the partials of the last chained weight gradient (no-op when nothing is pending)."""
    for key, st in list(_WG_CHAIN.items()):
        if device is not None and key != (device.index if device.index is not None else torch.cuda.current_device()):
            continue
        if st.pend.valid:
            L.check(L.load().bx_conv3x3_wgrad_finish(C.byref(st.pend), st.stream), "bx_conv3x3_wgrad_finish")
        st.keep = None


def _wgrad(x, dz, w: torch.Tensor, b: torch.Tensor, chain: bool = False):
    """chain=True: the caller promises to call wgrad_flush() before the gradients leave its hands."""
    lib = L.load()
    B, H, W, Cip = x.shape
    Co = dz.shape[3]
    dt = bx_dtype(x.dtype)
    algo = WGRAD_ALGO
    need = lib.bx_conv3x3_wgrad_workspace(B, H, W, Cip, Co, dt, algo)
    dw, db = new_grad(w), new_grad(b)
    if chain and WGRAD_CHAIN and algo != L.BX_ALGO_DIRECT:      # both storage types chain (shapes only the direct kernel covers finish the chain themselves)
        st = _wg_chain_state(x.device)
        if st.pend.valid and st.stream != _stream():
            wgrad_flush(x.device)                           # never carry a pending reduce across streams
        buf = st.ring[st.slot]
        if buf is None or buf.numel() < need:
            if buf is not None:
                # never free a partial buffer: kernels of an already captured hipGraph may point at it (an eager-era buffer
                # dropped while a later step was being captured was returned to the driver by the next capture's
                # empty_cache() -> memory access fault on replay)
                st.retired.append(buf)
            big = max([int(need), 1 << 20] + [b.numel() for b in st.ring if b is not None])
            buf = torch.empty(big, dtype=torch.uint8, device=x.device)
            st.ring[st.slot] = buf
        st.stream = _stream()
        with _Timed("wgrad", ("conv", B, H, W, Cip, Co)):
            L.check(lib.bx_conv3x3_wgrad_chained(_p(x), _p(dz), _p(dw), _p(db), B, H, W, w.shape[1], Cip, Co, dt, algo, _p(buf), buf.numel(),
                                                 C.byref(st.pend), st.stream), "bx_conv3x3_wgrad_chained")
        st.keep = (buf, dw, db)                             # what the pending reduce reads / writes
        st.slot ^= 1
        return dw, db
    ws = workspace(need, x.device)
    with _Timed("wgrad", ("conv", B, H, W, Cip, Co)):
        L.check(lib.bx_conv3x3_wgrad(_p(x), _p(dz), _p(dw), _p(db), B, H, W, w.shape[1], Cip, Co, dt, algo, _p(ws), ws.numel(), _stream()),
                "bx_conv3x3_wgrad")
    return dw, db


def _tail_desc(x, shape, dt, cfg, route=None) -> L.TailDesc:
    B, H, W, Cc = shape
    return L.TailDesc(B, H, W, x.shape[3], Cc, L.BX_POOL_MAX if cfg.pool == "max" else L.BX_POOL_AVG, 1 if cfg.training else 0,
                      cfg.eps, cfg.momentum, float(cfg.dropout_p), cfg.salt, bx_dtype(dt), _p(cfg.sync), _p(route))


class BlockFn(torch.autograd.Function):
    """relu(conv3x3) x3 -> pool -> BN -> dropout -> + conv1x1(bilinear(x))   (reference models.py:62-77).

    x: NHWC [B,H,W,Cin_p].  Returns NHWC [B,H/2,W/2,C].  With cfg.preact = k in {1,2,3} the k-th conv's
    PRE-ReLU output (what a hook on ``blockN.convK`` would see) is kept in cfg.capture["act"] and the
    gradient reaching it during backward in cfg.capture["grad"] (attribution targets inside a stage).
    """

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, bnw, bnb, w11, b11, rm, rv, nbt, cfg):
        lib = L.load()
        dt = x.dtype
        ws_, bs_ = (w1, w2, w3), (b1, b2, b3)
        acts, pre = [x], None
        B, H, W, _ = x.shape
        Cc = w3.shape[0]
        # conv3 with the pool and the batch statistics in its epilogue (two launches for conv3 + tail instead of four)
        fused = FUSE_POOL and dt == torch.bfloat16 and CONV_ALGO != L.BX_ALGO_DIRECT and cfg.preact != 3 and Cc >= 16
        packed3 = None
        first = 0
        masks = None
        # stage 1 (8 padded input channels -> 16 -> 16, bf16): conv1 and conv2 in one launch, conv1's output kept in LDS and written
        # out only when a backward pass (or a debugging hook) will read it
        if (CONV_PAIR and dt == torch.bfloat16 and CONV_ALGO != L.BX_ALGO_DIRECT and cfg.preact not in (1, 2)
                and lib.bx_conv3x3_pair_supported(x.shape[3], w1.shape[0], w2.shape[0], bx_dtype(dt))):
            pk = []
            for k in range(2):
                packed = cfg.prepacked.get(cfg.pack_base + k, False) if cfg.prepacked is not None else None
                pk.append(packed if packed is not None else _pack(ws_[k], flip=False, dtype=dt))
            if pk[0][1] is not None and pk[1][1] is not None:
                need_y1 = cfg.keep is not None or (getattr(cfg, "grad_mode", True) and any(ctx.needs_input_grad[:11]))
                y1 = torch.empty(B, H, W, w1.shape[0], dtype=dt, device=x.device) if need_y1 else None
                y2 = torch.empty(B, H, W, w2.shape[0], dtype=dt, device=x.device)
                if need_y1 and MASK_BITS and w1.shape[0] <= 32:     # the ReLU decisions of y1 / y2 as bits for the data gradients (1/8 of the
                    # bytes; stages 1-2: the later stages' data gradients are not bound by those bytes and read the activations)
                    masks = (torch.empty(B, H, W, w1.shape[0] // 4, dtype=torch.uint8, device=x.device),
                             torch.empty(B, H, W, w2.shape[0] // 4, dtype=torch.uint8, device=x.device))
                with _Timed("fwd", ("pair", B, H, W, x.shape[3], (w1.shape[0], w2.shape[0]))):
                    L.check(lib.bx_conv3x3_pair(_p(x), _p(pk[0][1]), _p(b1), _p(pk[1][1]), _p(b2), _p(y1), _p(y2),
                                                _p(masks[0]) if masks else None, _p(masks[1]) if masks else None, B, H, W, x.shape[3],
                                                w1.shape[0], w2.shape[0], bx_dtype(dt), _stream()), "bx_conv3x3_pair")
                acts += [y1 if y1 is not None else x.new_empty(0), y2]
                first = 2
        for k in range(first, 3):
            packed = cfg.prepacked.get(cfg.pack_base + k, False) if cfg.prepacked is not None else None
            if packed is None:
                packed = _pack(ws_[k], flip=False, dtype=dt)
            if k == 2 and fused and packed[1] is not None:
                packed3 = packed
                break
            if cfg.preact == k + 1:
                pre = _conv(acts[-1], packed, bs_[k], None, None, False, dt)
                y = torch.empty_like(pre)
                L.check(lib.bx_relu(_p(pre), _p(y), pre.numel(), bx_dtype(dt), _stream()), "bx_relu")
            else:
                y = _conv(acts[-1], packed, bs_[k], None, None, True, dt)
            acts.append(y)
        seed = None
        if cfg.training and cfg.dropout_p > 0:
            seed = cfg.seed if cfg.seed is not None else next_seed(x.device)       # the model hands one seed to all stages (salts differ)
        pooled = torch.empty(B, H // 2, W // 2, Cc, dtype=dt, device=x.device)
        out = torch.empty_like(pooled)
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        route = None
        if packed3 is not None:
            # What a backward pass needs of conv3's output is only where each pooled element's gradient goes: the fused launch writes
            # that as one nibble per pooled element (bxTailDesc.route) and conv3's full-resolution output is not stored at all --
            # unless a debugging hook asked for the activations.  Without a backward in sight (Grad-CAM sweeps, inference): neither.
            need_bwd = getattr(cfg, "grad_mode", True) and any(ctx.needs_input_grad[:11])
            use_route = TAIL_ROUTE and cfg.keep is None
            y3 = torch.empty(B, H, W, Cc, dtype=dt, device=x.device) if (cfg.keep is not None or (need_bwd and not use_route)) else None
            if need_bwd and use_route:
                route = torch.empty(B * (H // 2) * (W // 2) * Cc // 2, dtype=torch.uint8, device=x.device)
            desc = _tail_desc(x, (B, H, W, Cc), dt, cfg, route)
            ws = workspace(lib.bx_block_tail_workspace(C.byref(desc)), x.device)
            if CONV_PROFILE is not None:                    # the library records the pair around the convolution kernel of this call
                pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                pe0.record(); pe1.record()                  # (creates the hipEvent handles)
                L.check(lib.bx_profile_next_conv3(pe0.cuda_event, pe1.cuda_event), "bx_profile_next_conv3")
                CONV_PROFILE.append(("fwd", pe0, pe1, ("conv3+pool", B, H, W, acts[2].shape[3], Cc)))
            L.check(lib.bx_block_conv3_tail_fwd(C.byref(desc), _p(acts[2]), _p(packed3[1]), _p(b3), _p(y3), _p(x), _p(w11), w11.shape[1], _p(b11),
                                                _p(bnw), _p(bnb), _p(rm), _p(rv), _p(nbt), _p(seed), _p(pooled), _p(out), _p(mean), _p(invstd),
                                                _p(ws), ws.numel(), _stream()), "bx_block_conv3_tail_fwd")
            acts.append(y3)
        else:
            y3 = acts[3]
            desc = _tail_desc(x, tuple(y3.shape), dt, cfg)
            ws = workspace(lib.bx_block_tail_workspace(C.byref(desc)), x.device)
            L.check(lib.bx_block_tail_fwd(C.byref(desc), _p(y3), _p(x), _p(w11), w11.shape[1], _p(b11), _p(bnw), _p(bnb), _p(rm), _p(rv), _p(nbt),
                                          _p(seed), _p(pooled), _p(out), _p(mean), _p(invstd), _p(ws), ws.numel(), _stream()), "bx_block_tail_fwd")
        ctx.cfg, ctx.desc, ctx.seed = cfg, desc, seed
        ctx.masks = masks                                   # (plain attributes: uint8 side outputs of the pair launch, never differentiated)
        ctx.route, ctx.y3_shape = route, (B, H, W, Cc)
        ctx.save_for_backward(x, acts[1], acts[2], y3 if y3 is not None else x.new_empty(0), pooled, mean, invstd, w1, b1, w2, b2, w3, b3,
                              bnw, bnb, w11, b11)
        if pre is not None:
            cfg.capture["act"] = pre
        if cfg.keep is not None:                            # debugging / parity tooling: see keep_block_activations()
            cfg.keep["acts"] = (acts[1], acts[2], y3)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        x, y1, y2, y3, pooled, mean, invstd, w1, b1, w2, b2, w3, b3, bnw, bnb, w11, b11 = ctx.saved_tensors
        cfg, desc, seed = ctx.cfg, ctx.desc, ctx.seed
        dt = x.dtype
        need_dx = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[1:11])
        dout = dout.contiguous()
        dz3 = torch.empty(ctx.y3_shape, dtype=dt, device=x.device)
        if y3.numel() == 0:
            y3 = None                                       # not stored: the route nibbles of the descriptor stand in for it
        dx_skip = torch.empty_like(x) if need_dx else None
        # arena slices only for gradients autograd asked for (a frozen attribution pass must not touch the trainer's arena)
        need_bn = ctx.needs_input_grad[7] or ctx.needs_input_grad[8]
        d_bnw, d_bnb = (new_grad(bnw), new_grad(bnb)) if need_bn else (torch.empty_like(bnw), torch.empty_like(bnb))
        d_w11, d_b11 = (new_grad(w11), new_grad(b11)) if need_w else (None, None)
        ws = workspace(lib.bx_block_tail_workspace(C.byref(desc)), x.device)
        L.check(lib.bx_block_tail_bwd(C.byref(desc), _p(dout), _p(y3), _p(x), _p(pooled), _p(w11), w11.shape[1], _p(bnw), _p(mean), _p(invstd),
                                      _p(seed), _p(dz3), _p(dx_skip), _p(d_bnw), _p(d_bnb), _p(d_w11), _p(d_b11), _p(ws), ws.numel(), _stream()),
                "bx_block_tail_bwd")
        if cfg.preact == 3:
            cfg.capture["grad"] = dz3
        acts = (x, y1, y2)
        wts = (w1, w2, w3)
        bss = (b1, b2, b3)
        grads_w, grads_b = [None] * 3, [None] * 3
        dz = dz3
        use_side = OVERLAP and need_w and CONV_PROFILE is None
        side = side_stream("wgrad", x.device) if use_side else None
        if use_side:
            _join_after_backward()
        for k in (2, 1, 0):
            if need_w and side is not None:
                # dW/db of this layer only feed the optimizer: compute them beside the data-gradient chain
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    grads_w[k], grads_b[k] = _wgrad(acts[k], dz, wts[k], bss[k])
                dz.record_stream(side); acts[k].record_stream(side)
            elif need_w:
                grads_w[k], grads_b[k] = _wgrad(acts[k], dz, wts[k], bss[k], chain=True)
            packed = cfg.prepacked.get(cfg.pack_base + k, True) if cfg.prepacked is not None else None
            if k > 0:
                mb = ctx.masks is not None and CONV_ALGO != L.BX_ALGO_DIRECT
                dz = _conv(dz, packed or _pack(wts[k], flip=True, dtype=dt), None, ctx.masks[k - 1] if mb else acts[k], None, False, dt,
                           mask_bits=mb)
                if cfg.preact == k:
                    cfg.capture["grad"] = dz
            elif need_dx:
                dz = _conv(dz, packed or _pack(wts[0], flip=True, dtype=dt), None, None, dx_skip, False, dt, carry=need_w and side is None)
        if need_w and side is None:
            wgrad_flush(x.device)                           # conv1's partial sum: nothing may be pending when the gradients are returned
        dx = dz if need_dx else None
        if not need_bn:
            d_bnw = d_bnb = None
        return (dx, grads_w[0], grads_b[0], grads_w[1], grads_b[1], grads_w[2], grads_b[2], d_bnw, d_bnb, d_w11, d_b11,
                None, None, None, None)


class GapFcLsmFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) -> Linear -> LogSoftmax over an NHWC feature map (reference models.py:103-106)."""

    @staticmethod
    def forward(ctx, feat, w, b):
        B, H, W, Cc = feat.shape
        N = w.shape[0]
        gap = torch.empty(B, Cc, dtype=torch.float32, device=feat.device)
        logp = torch.empty(B, N, dtype=torch.float32, device=feat.device)
        L.check(L.load().bx_gap_fc_lsm_fwd(_p(feat), _p(w), _p(b), _p(gap), _p(logp), B, H * W, Cc, N, bx_dtype(feat.dtype), _stream()),
                "bx_gap_fc_lsm_fwd")
        ctx.shape, ctx.dt = (B, H, W, Cc), feat.dtype
        ctx.save_for_backward(gap, logp, w, b)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        gap, logp, w, b = ctx.saved_tensors
        B, H, W, Cc = ctx.shape
        N = w.shape[0]
        dfeat = torch.empty(B, H, W, Cc, dtype=ctx.dt, device=gap.device) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw, db = (new_grad(w), new_grad(b)) if need_w else (None, None)
        dlogp = dlogp.contiguous()          # keep any temporary alive across the launch
        L.check(L.load().bx_gap_fc_lsm_bwd(_p(dlogp), _p(logp), _p(gap), _p(w), _p(dfeat), _p(dw), _p(db), B, H * W, Cc, N,
                                           bx_dtype(ctx.dt), _stream()), "bx_gap_fc_lsm_bwd")
        return dfeat, dw, db


class LinearLsmFn(torch.autograd.Function):
    """Flatten -> Linear -> LogSoftmax on fp32 features (reference models.py:286-288)."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, K = x.shape
        N = w.shape[0]
        logp = torch.empty(B, N, dtype=torch.float32, device=x.device)
        L.check(L.load().bx_linear_lsm_fwd(_p(x), _p(w), _p(b), _p(logp), B, K, N, _stream()), "bx_linear_lsm_fwd")
        ctx.save_for_backward(x, logp, w, b)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        x, logp, w, b = ctx.saved_tensors
        B, K = x.shape
        N = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw, db = (new_grad(w), new_grad(b)) if need_w else (None, None)
        dlogp = dlogp.contiguous()
        L.check(L.load().bx_linear_lsm_bwd(_p(dlogp), _p(logp), _p(x), _p(w), _p(dx), _p(dw), _p(db), B, K, N, _stream()),
                "bx_linear_lsm_bwd")
        return dx, dw, db


class FusionHeadFn(torch.autograd.Function):
    """cat -> Linear(2N,Hd) -> ReLU -> Linear(Hd,N) -> LogSoftmax (reference XAI_Multimodality.py:1095-1105)."""

    @staticmethod
    def forward(ctx, e, s, w1, b1, w2, b2):
        B, N = e.shape
        Hd = w1.shape[0]
        hidden = torch.empty(B, Hd, dtype=torch.float32, device=e.device)
        logp = torch.empty(B, N, dtype=torch.float32, device=e.device)
        e, s = e.contiguous(), s.contiguous()
        L.check(L.load().bx_fusion_head_fwd(_p(e), _p(s), _p(w1), _p(b1), _p(w2), _p(b2), _p(hidden), _p(logp), B, N, Hd, _stream()),
                "bx_fusion_head_fwd")
        ctx.save_for_backward(e, s, hidden, logp, w1, b1, w2, b2)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        e, s, hidden, logp, w1, b1, w2, b2 = ctx.saved_tensors
        B, N = e.shape
        Hd = w1.shape[0]
        de = torch.empty_like(e) if ctx.needs_input_grad[0] else None
        ds = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        need_w = any(ctx.needs_input_grad[2:])
        dw1, db1, dw2, db2 = (new_grad(w1), new_grad(b1), new_grad(w2), new_grad(b2)) if need_w else (None,) * 4
        dlogp = dlogp.contiguous()
        L.check(L.load().bx_fusion_head_bwd(_p(dlogp), _p(logp), _p(hidden), _p(e), _p(s), _p(w1), _p(w2), _p(de), _p(ds),
                                            _p(dw1), _p(db1), _p(dw2), _p(db2), B, N, Hd, _stream()), "bx_fusion_head_bwd")
        return de, ds, dw1, db1, dw2, db2


class MultimodalHeadFn(torch.autograd.Function):
    """GAP+fc+LogSoftmax (spectrogram), dense+LogSoftmax (EEG) and the fusion head in one launch forward, two backward
    (reference models.py:103-106, :286-288, XAI_Multimodality.py:1095-1105).  feat: NHWC block5 map, efeat fp32 [B,K]."""

    @staticmethod
    def forward(ctx, feat, efeat, fcw, fcb, dw, db, w1, b1, w2, b2):
        B, H, W, Cc = feat.shape
        K, N, Hd = efeat.shape[1], fcw.shape[0], w1.shape[0]
        dev = feat.device
        efeat = efeat.contiguous()
        gap = torch.empty(B, Cc, dtype=torch.float32, device=dev)
        slp, elp, logp = (torch.empty(B, N, dtype=torch.float32, device=dev) for _ in range(3))
        hidden = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        L.check(L.load().bx_mm_head_fwd(_p(feat), _p(efeat), _p(fcw), _p(fcb), _p(dw), _p(db), _p(w1), _p(b1), _p(w2), _p(b2), _p(gap), _p(slp),
                                        _p(elp), _p(hidden), _p(logp), B, H * W, Cc, K, N, Hd, bx_dtype(feat.dtype), _stream()), "bx_mm_head_fwd")
        ctx.shape, ctx.dt = (B, H, W, Cc), feat.dtype
        ctx.save_for_backward(gap, slp, elp, hidden, logp, efeat, fcw, fcb, dw, db, w1, b1, w2, b2)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        gap, slp, elp, hidden, logp, efeat, fcw, fcb, dw, db, w1, b1, w2, b2 = ctx.saved_tensors
        B, H, W, Cc = ctx.shape
        K, N, Hd = efeat.shape[1], fcw.shape[0], w1.shape[0]
        lib = L.load()
        dfeat = torch.empty(B, H, W, Cc, dtype=ctx.dt, device=gap.device) if ctx.needs_input_grad[0] else None
        defeat = torch.empty_like(efeat) if ctx.needs_input_grad[1] else None
        gl = [new_grad(t) if need else None for t, need in zip((fcw, fcb, dw, db, w1, b1, w2, b2), ctx.needs_input_grad[2:])]
        ws = workspace(lib.bx_mm_head_workspace(B, N, Hd), gap.device)
        dlogp = dlogp.contiguous()
        L.check(lib.bx_mm_head_bwd(_p(dlogp), _p(logp), _p(hidden), _p(slp), _p(elp), _p(gap), _p(efeat), _p(fcw), _p(dw), _p(w1), _p(w2),
                                   _p(dfeat), _p(defeat), *[_p(t) for t in gl], _p(ws), ws.numel(), B, H * W, Cc, K, N, Hd, bx_dtype(ctx.dt),
                                   _stream()), "bx_mm_head_bwd")
        return (dfeat, defeat, *gl)


_REDUCTIONS = {"mean": 0, "batchmean": 1, "sum": 2}


_ONES = {}


def unit_gradient(device) -> torch.Tensor:
    """A cached scalar 1.0 on ``device`` for ``loss.backward(gradient=...)``: autograd would otherwise fill a fresh
    ones-tensor every step (a launch), and KLDivFn recognises this very tensor and skips its multiply (another launch)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _ONES.get(key)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _ONES[key] = t
    return t


class KLDivFn(torch.autograd.Function):
    """nn.KLDivLoss(reduction)(log_probs, target)  (reference XAI_Multimodality.py:1989,1599)."""

    @staticmethod
    def forward(ctx, logp, target, reduction, grad_scale):
        B, N = logp.shape
        loss = torch.empty((), dtype=torch.float32, device=logp.device)
        dlogp = torch.empty_like(logp)
        logp_c, target_c = logp.contiguous(), target.contiguous().float()
        L.check(L.load().bx_kldiv_fwd_bwd(_p(logp_c), _p(target_c), _p(loss), _p(dlogp), B, N,
                                          _REDUCTIONS[reduction], float(grad_scale), _stream()), "bx_kldiv_fwd_bwd")
        ctx.save_for_backward(dlogp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogp,) = ctx.saved_tensors
        one = _ONES.get(dlogp.device.index if dlogp.device.index is not None else torch.cuda.current_device())
        if one is not None and g.data_ptr() == one.data_ptr():
            return dlogp, None, None, None                  # upstream gradient is the cached constant 1: nothing to multiply
        out = torch.empty_like(dlogp)
        g = g.contiguous()
        L.check(L.load().bx_scale_dev(_p(dlogp), _p(g), _p(out), dlogp.numel(), _stream()), "bx_scale_dev")
        return out, None, None, None


class EegFeaturesFn(torch.autograd.Function):
    """EEGNet up to Flatten (reference models.py:271-285). x fp32 [B,1,Chans,T] -> feat fp32 [B, F2*(T//32)]."""

    @staticmethod
    def forward(ctx, x, c1w, bn1w, bn1b, dww, bn2w, bn2b, sepw, bn3w, bn3b, bufs, cfg):
        lib = L.load()
        _require_gpu(x, "eeg input")
        B, _, Ch, T = x.shape
        x = x.contiguous().float()
        # collapsed front end (no conv1 output tensor).  Training: only when nobody can ask for the input's gradient.  Evaluation
        # mode (BatchNorm1 a fixed affine map): whenever no PARAMETER needs a gradient -- inference, Grad-CAM sweeps, saliency /
        # integrated-gradients passes (the input gradient has a collapsed form too)
        params_need_grad = getattr(cfg, "grad_mode", True) and any(ctx.needs_input_grad[1:10])      # under no_grad nobody does
        collapse = 1 if EEG_COLLAPSE and ((cfg.training and not ctx.needs_input_grad[0]) or
                                          (not cfg.training and not params_need_grad)) else 0
        desc = L.EegDesc(B, Ch, T, cfg.F1, cfg.D, cfg.F2, cfg.K1, cfg.K2, cfg.P1, cfg.P2, 1 if cfg.training else 0, cfg.eps, cfg.momentum,
                         float(cfg.dropout_p), cfg.salt, bx_dtype(cfg.dtype), collapse, float(getattr(cfg, "dropout_p2", -1.0)))
        if INPUT_SLOTS and collapse and not cfg.training and not torch.is_grad_enabled() and cfg.K1 == 64:
            desc.x_slot = _take_slot(x)                      # a graph input read through its device slot (GradCamSweep)
        nsaved = lib.bx_eeg_saved_bytes(C.byref(desc))
        if nsaved == 0:
            raise RuntimeError("brainxai: EEGNet geometry outside the kernels' range (F1*D, F2 <= 1024, kernel lengths <= 4096, tensors below 2^31 elements)")
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        ws = workspace(lib.bx_eeg_workspace(C.byref(desc)), x.device)
        params = L.EegParams(_p(c1w), _p(bn1w), _p(bn1b), _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(dww), _p(bn2w), _p(bn2b), _p(bufs[3]),
                             _p(bufs[4]), _p(bufs[5]), _p(sepw), _p(bn3w), _p(bn3b), _p(bufs[6]), _p(bufs[7]), _p(bufs[8]))
        T2 = (T // cfg.P1) // cfg.P2
        feat = torch.empty(B, cfg.F2 * T2, dtype=torch.float32, device=x.device)
        seed = None
        if cfg.training and (cfg.dropout_p > 0 or getattr(cfg, "dropout_p2", 0.0) > 0):
            seed = getattr(cfg, "seed", None)
            if seed is None:
                seed = next_seed(x.device, "eeg")
        L.check(lib.bx_eeg_features_fwd(C.byref(desc), C.byref(params), _p(x), _p(seed), _p(feat), _p(saved), _p(ws), ws.numel(), _stream()),
                "bx_eeg_features_fwd")
        ctx.desc, ctx.params, ctx.seed, ctx.bufs = desc, params, seed, bufs
        ctx.save_for_backward(x, saved, c1w, bn1w, bn1b, dww, bn2w, bn2b, sepw, bn3w, bn3b)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = L.load()
        if OVERLAP_EEG:
            dfeat.record_stream(torch.cuda.current_stream())   # produced on the main stream, read here on the branch's stream
        x, saved, c1w, bn1w, bn1b, dww, bn2w, bn2b, sepw, bn3w, bn3b = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[1:10])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gl = [new_grad(t) for t in (c1w, bn1w, bn1b, dww, bn2w, bn2b, sepw, bn3w, bn3b)] if need_w else [None] * 9
        grads = L.EegGrads(*[_p(t) for t in gl])
        ws = workspace(lib.bx_eeg_workspace(C.byref(ctx.desc)), x.device)
        dfeat = dfeat.contiguous()
        L.check(lib.bx_eeg_features_bwd(C.byref(ctx.desc), C.byref(ctx.params), _p(x), _p(dfeat), _p(ctx.seed), _p(saved),
                                        C.byref(grads), _p(dx), _p(ws), ws.numel(), _stream()), "bx_eeg_features_bwd")
        return (dx, *gl, None, None)


class EegDeepHeadFn(torch.autograd.Function):
    """EEGNetAttentionDeep after block 2 (reference models.py:215-233 + Attention :109-134): conv2 -> BN4 -> ELU -> pool8 ->
    dropout -> attention over time -> dense1 -> dense2 -> LogSoftmax.  feat fp32 [B, 16*T2] -> (logp [B,N], attn [B,L,L])."""

    @staticmethod
    def forward(ctx, feat, c2w, bn4w, bn4b, wq, bq, wk, bk, wv, bv, w1, b1, w2, b2, bufs, cfg):
        lib = L.load()
        _require_gpu(feat, "EEG features")
        B = feat.shape[0]
        feat = feat.contiguous().float()
        desc = L.EegDeepDesc(B, cfg.T2, cfg.F2, cfg.F3, cfg.K3, cfg.P3, w1.shape[0], w2.shape[0], 1 if cfg.training else 0, cfg.eps,
                             cfg.momentum, float(cfg.dropout_p), cfg.salt)
        nsaved = lib.bx_eeg_deep_saved_bytes(C.byref(desc))
        if nsaved == 0:
            raise RuntimeError("brainxai: unsupported EEGNetAttentionDeep geometry (needs F2=16, F3=32, 8 <= Samples//32, Samples//256 <= 32, "
                               "dense1 width a power of two in [32,256], <= 16 classes)")
        if w1.data_ptr() % 16:      # the kernels read dense1.weight 16 bytes at a time; a packed parameter arena may misalign it
            w1 = w1.clone()
        L_ = cfg.T2 // cfg.P3
        saved = torch.empty(nsaved, dtype=torch.uint8, device=feat.device)
        ws = workspace(lib.bx_eeg_deep_workspace(C.byref(desc)), feat.device)
        params = L.EegDeepParams(_p(c2w), _p(bn4w), _p(bn4b), _p(bufs[0]), _p(bufs[1]), _p(bufs[2]), _p(wq), _p(bq), _p(wk), _p(bk),
                                 _p(wv), _p(bv), _p(w1), _p(b1), _p(w2), _p(b2))
        logp = torch.empty(B, w2.shape[0], dtype=torch.float32, device=feat.device)
        attn = torch.empty(B, L_, L_, dtype=torch.float32, device=feat.device)
        seed = next_seed(feat.device, "eeg") if (cfg.training and cfg.dropout_p > 0) else None
        L.check(lib.bx_eeg_deep_fwd(C.byref(desc), C.byref(params), _p(feat), _p(seed), _p(logp), _p(attn), _p(saved), _p(ws), ws.numel(),
                                    _stream()), "bx_eeg_deep_fwd")
        ctx.desc, ctx.params, ctx.seed, ctx.bufs = desc, params, seed, bufs
        ctx.save_for_backward(feat, saved, attn, c2w, bn4w, bn4b, wq, bq, wk, bk, wv, bv, w1, b1, w2, b2)
        ctx.mark_non_differentiable(attn)
        return logp, attn

    @staticmethod
    def backward(ctx, dlogp, _dattn):
        lib = L.load()
        feat, saved, attn, *plist = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[1:14])
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        gl = [new_grad(t) for t in plist] if need_w else [None] * 13
        grads = L.EegDeepGrads(*[_p(t) for t in gl])
        ws = workspace(lib.bx_eeg_deep_workspace(C.byref(ctx.desc)), feat.device)
        dlogp = dlogp.contiguous()
        L.check(lib.bx_eeg_deep_bwd(C.byref(ctx.desc), C.byref(ctx.params), _p(feat), _p(dlogp), _p(attn), _p(ctx.seed), _p(saved),
                                    C.byref(grads) if need_w else None, _p(dfeat), _p(ws), ws.numel(), _stream()), "bx_eeg_deep_bwd")
        return (dfeat, *gl, None, None)


class AttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(d)) V with Linear query/key/value maps (reference models.py:117-134): x [B,L,32] -> (out, weights)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv):
        lib = L.load()
        _require_gpu(x, "attention input")
        if x.dim() != 3 or x.shape[2] != 32 or x.shape[1] > 32 or tuple(wq.shape) != (32, 32):
            raise RuntimeError("brainxai Attention: needs tokens [B, L <= 32, 32] and input_dim = attention_dim = 32")
        B, L_, D = x.shape
        x = x.contiguous().float()
        out = torch.empty(B, L_, D, dtype=torch.float32, device=x.device)
        attn = torch.empty(B, L_, L_, dtype=torch.float32, device=x.device)
        qkv = torch.empty(B, 3, L_, D, dtype=torch.float32, device=x.device)
        L.check(lib.bx_attention_fwd(_p(x), _p(wq), _p(bq), _p(wk), _p(bk), _p(wv), _p(bv), _p(out), _p(attn), _p(qkv), B, L_, D, _stream()),
                "bx_attention_fwd")
        ctx.save_for_backward(x, attn, qkv, wq, bq, wk, bk, wv, bv)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        lib = L.load()
        x, attn, qkv, wq, bq, wk, bk, wv, bv = ctx.saved_tensors
        B, L_, D = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gl = [new_grad(t) if need else None for t, need in zip((wq, bq, wk, bk, wv, bv), ctx.needs_input_grad[1:])]
        ws = workspace(lib.bx_attention_workspace(B), x.device)
        dout = dout.contiguous()
        dattn = dattn.contiguous() if dattn is not None else None
        L.check(lib.bx_attention_bwd(_p(dout), _p(dattn), _p(x), _p(attn), _p(qkv), _p(wq), _p(wk), _p(wv), _p(dx), *[_p(t) for t in gl],
                                     _p(ws), ws.numel(), B, L_, D, _stream()), "bx_attention_bwd")
        return (dx, *gl)


def block_cfg(**kw) -> SimpleNamespace:
    base = dict(pool="max", training=False, dropout_p=0.0, eps=1e-5, momentum=0.1, salt=0, preact=0, capture=None,
                prepacked=None, pack_base=0, seed=None, keep=None, sync=None)
    base.update(kw)
    return SimpleNamespace(**base)


def keep_block_activations(model, on=True):
    """Parity tooling: make every Block of ``model`` keep the post-ReLU outputs of its three convolutions from its next
    forward (they are saved for backward anyway; this only exposes them).  Returns {module name: dict}; after a forward
    ``dict["acts"]`` holds (y1, y2, y3) as channels-last tensors [B,H,W,C].  ``on=False`` switches it off again."""
    out = {}
    for name, mod in model.named_modules():
        if hasattr(mod, "_keep"):
            mod._keep = {} if on else None
            if on:
                out[name] = mod._keep
    return out
