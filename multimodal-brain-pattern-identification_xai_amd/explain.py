"""Attribution over the multimodal model on the HIP path.

* ``grad_cam``                -- canonical Grad-CAM (the reference ships none): forward,
                                 backward from the class score to the target layer only, then the fused
                                 activation x gradient channel-reduce kernel (bx_gradcam_reduce).
* ``saliency`` / ``generate_saliency_maps`` -- reference XAI_Multimodality.py:3101-3133.
* ``integrated_gradients``    -- Captum-default semantics (imported but never called by the reference, NB:51).
"""
from __future__ import annotations

import contextlib
import re

import numpy as np
import torch

from . import _lib as L
from . import ops
from .ops import _p, _stream


@contextlib.contextmanager
def _eval_frozen(model):
    """eval() mode with parameters detached from autograd (so backward passes skip every weight-gradient
    kernel); both restored on exit."""
    was_training = model.training
    flags = [(p, p.requires_grad) for p in model.parameters()]
    model.eval()
    for p, _ in flags:
        p.requires_grad_(False)
    try:
        with ops.pack_reuse():                            # the weights stand for the whole pass: packed once, on its first forward
            yield
    finally:
        for p, f in flags:
            p.requires_grad_(f)
        model.train(was_training)


def _resolve(model, dotted):
    mod = model
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def _class_seed(out: torch.Tensor, class_mode: int, rows=None) -> torch.Tensor:
    """[rows, n_classes] one-hot gradient seeds: row r selects class ``class_mode`` (>= 0) or the arg-max class of
    ``out[r % B]`` (-1); one library launch (bx_class_seed)."""
    B, n = out.shape
    rows = B if rows is None else rows
    seed = torch.empty(rows, n, dtype=torch.float32, device=out.device)
    logp = out.detach().float().contiguous()
    L.check(L.load().bx_class_seed(_p(logp), _p(seed), rows, B, n, int(class_mode), _stream()), "bx_class_seed")
    return seed


def _reduce(A_nhwc, G_nhwc, maps_per_act, relu):
    lib = L.load()
    n_maps, h, w, c = G_nhwc.shape
    cam = torch.empty(n_maps, h, w, dtype=torch.float32, device=G_nhwc.device)
    wts = torch.empty(n_maps, c, dtype=torch.float32, device=G_nhwc.device)
    L.check(lib.bx_gradcam_reduce(_p(A_nhwc), _p(G_nhwc), _p(cam), _p(wts), n_maps, maps_per_act, h * w, c, 1 if relu else 0,
                                  ops.bx_dtype(A_nhwc.dtype), _stream()), "bx_gradcam_reduce")
    return cam, wts


def resize_bilinear(maps: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(maps[:,None], size, mode='bilinear', align_corners=False)[:,0] on fp32 [N,h,w]."""
    n, h, w = maps.shape
    H, W = size
    maps = maps.contiguous()
    out = torch.empty(n, H, W, dtype=torch.float32, device=maps.device)
    L.check(L.load().bx_resize_bilinear(_p(maps), _p(out), n, h, w, H, W, _stream()), "bx_resize_bilinear")
    return out


_TARGET = re.compile(r"^(?:spectrogram_model\.)?block([1-5])(?:\.conv([1-3]))?$")


def _grad_cam_last_stage(model, eeg, spec, class_idx, upsample, relu, return_parts):
    """Default target (the last stage feeds the heads directly): no autograd and no framework arithmetic at all.  The two
    branches run forward, then ONE launch (bx_gradcam_head) does both heads forward, their backward for every requested class
    and the activation x gradient channel reduce; a second launch upsamples the maps."""
    lib = L.load()
    sm = model.spectrogram_model
    if class_idx is None:
        mode = -1
    elif isinstance(class_idx, str):
        if class_idx != "all":
            raise ValueError(class_idx)
        mode = -2
    else:
        mode = int(class_idx)
    with torch.no_grad():
        em = model.eeg_model
        H, W = spec.shape[-2:]
        if upsample and not return_parts and W % 4 == 0 and hasattr(model, "_fusable") and model._fusable():
            # sweep form: the EEG branch's dense + LogSoftmax and the up-sampling ride in the head launch (two launches fewer per batch)
            # (the EEG branch beside the spectrogram branch, as the captured training step runs them, does not pay here: 440 vs 433 us
            # per batch -- the evaluation-mode branch is 55 us of five launches and the fork / join costs about as much as it hides)
            ef = em.features(eeg).contiguous()
            A = sm.features(spec).permute(0, 2, 3, 1).contiguous()
            B, h, w, C = A.shape
            N, Hd = model.fc2.out_features, model.fc1.out_features
            nm = N if mode == -2 else 1
            if ef.shape[1] != em.dense.in_features:
                raise RuntimeError(f"EEGNet: {ef.shape[1]} features but dense expects {em.dense.in_features} (Samples mismatch)")
            lds_floats = 2 * C + 2 * Hd + 5 * N + (256 // (C // 8)) * C + h * w
            if lds_floats * 4 <= 64 * 1024:
                maps = torch.empty(B * nm, H, W, dtype=torch.float32, device=A.device)
                L.check(lib.bx_gradcam_head_sweep(_p(A), _p(ef), _p(em.dense.weight), _p(em.dense.bias), ef.shape[1], _p(sm.fc.weight),
                                                  _p(sm.fc.bias), _p(model.fc1.weight), _p(model.fc1.bias), _p(model.fc2.weight),
                                                  _p(model.fc2.bias), None, _p(maps), B, h, w, C, N, Hd, H, W, mode, 1 if relu else 0,
                                                  ops.bx_dtype(A.dtype), _stream()), "bx_gradcam_head_sweep")
                return maps.reshape(B, nm, H, W) if isinstance(class_idx, str) else maps
            e = ops.LinearLsmFn.apply(ef, em.dense.weight, em.dense.bias).contiguous()
        else:
            e = em(eeg).contiguous()
            A = sm.features(spec).permute(0, 2, 3, 1).contiguous()
        B, h, w, C = A.shape
        N, Hd = model.fc2.out_features, model.fc1.out_features
        nm = N if mode == -2 else 1
        dev = A.device
        out = torch.empty(B, N, dtype=torch.float32, device=dev)
        cam = torch.empty(B * nm, h, w, dtype=torch.float32, device=dev)
        raw = torch.empty_like(cam) if return_parts else None
        wts = torch.empty(B * nm, C, dtype=torch.float32, device=dev) if return_parts else None
        L.check(lib.bx_gradcam_head(_p(A), _p(e), _p(sm.fc.weight), _p(sm.fc.bias), _p(model.fc1.weight), _p(model.fc1.bias),
                                    _p(model.fc2.weight), _p(model.fc2.bias), _p(out), _p(cam), _p(raw), _p(wts), B, h * w, C, N, Hd, mode,
                                    1 if relu else 0, ops.bx_dtype(A.dtype), _stream()), "bx_gradcam_head")
        if upsample:
            cam = resize_bilinear(cam, spec.shape[-2:])
    stacked = isinstance(class_idx, str)
    shape = (lambda t: t.reshape(B, nm, *t.shape[1:])) if stacked else (lambda t: t)
    if return_parts:
        return shape(cam), shape(raw), shape(wts), A, out
    return shape(cam)


class GradCamSweep:
    """Grad-CAM at the default target replayed from captured hipGraphs -- for sweeps over many batches (BASELINE configs[3]:
    10 000 samples, all classes).  The launches of `grad_cam` are captured once per batch shape on static input buffers (the
    ragged last batch of a sweep gets its own capture the first time it is seen); a call copies the batch in and replays them,
    so the sweep runs at GPU speed instead of at the host's launch rate.  The returned tensor is that graph's static output
    buffer: clone it if it must outlive the next call with the same shape.

    The captured launches hold no weight-packing jobs: a call re-packs first when a parameter changed since the last pack -- as far
    as torch's version counters and this library's own optimizer kernels can tell.  A parameter rewritten through a ``.data`` view
    between two calls is NOT seen: call ``invalidate()`` after such an edit.

        sweep = GradCamSweep(model, eeg_batch, spec_batch, class_idx="all")
        for eeg, spec in loader:
            maps = sweep(eeg, spec)          # [B, 6, H, W]
    """

    def __init__(self, model, eeg, spec, class_idx="all", upsample=True, relu=True):
        if not (eeg.is_cuda and spec.is_cuda):
            raise RuntimeError("brainxai.GradCamSweep needs CUDA tensors; there is no CPU path")
        self.model, self.args = model, (class_idx, upsample, relu)
        self._graphs = {}
        self._capture(eeg, spec)

    def invalidate(self):
        """Force the next call to re-pack the weights (after parameter edits torch cannot see, e.g. through ``.data``)."""
        ops.bump_param_epoch()

    @staticmethod
    def _slot_ready(t):
        return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()

    def _capture(self, eeg, spec):
        model = self.model
        class_idx, upsample, relu = self.args
        s_eeg, s_spec = eeg.detach().clone().contiguous(), spec.detach().clone().contiguous()
        # The kernels that read the raw batch take its address from a device slot (ops.INPUT_SLOTS): a replay on the caller's own
        # fp32 tensors costs one 16-byte store instead of two copies (43 MB per batch of 64 at the bench shape, ~20 us).  Inputs in
        # another dtype / layout still go through the static buffers.
        slots = torch.zeros(2, dtype=torch.int64, device=s_eeg.device) if self._slot_ready(s_eeg) and self._slot_ready(s_spec) else None
        if slots is not None:
            L.check(L.load().bx_store_u64x2(slots.data_ptr(), s_eeg.data_ptr(), s_spec.data_ptr(), ops._stream()), "bx_store_u64x2")
            ops.INPUT_SLOTS[s_eeg.data_ptr()] = slots.data_ptr()
            ops.INPUT_SLOTS[s_spec.data_ptr()] = slots.data_ptr() + 8
        was_training = model.training
        model.eval()
        try:
            with ops.pack_reuse():                       # (bumps the parameter epoch: the first warm-up run packs the weights)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):                   # allocate workspaces / pack tables on the capture stream
                        _grad_cam_last_stage(model, s_eeg, s_spec, class_idx, upsample, relu, False)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = _grad_cam_last_stage(model, s_eeg, s_spec, class_idx, upsample, relu, False)
        finally:
            model.train(was_training)
            used = (False, False)
            if slots is not None:
                # which inputs the captured kernels really read through their slot (fp32 storage converts the spectrogram in another
                # kernel, a non-default EEGNet geometry reads x in several): the others are fed through the static buffers
                used = (ops.INPUT_SLOTS.pop(s_eeg.data_ptr(), None) == "used", ops.INPUT_SLOTS.pop(s_spec.data_ptr(), None) == "used")
                ops._SLOT_ADDR.pop(s_eeg.data_ptr(), None)
                ops._SLOT_ADDR.pop(s_spec.data_ptr(), None)
        plan = getattr(model.spectrogram_model, "_pack_plan", None)
        # the warm-up runs packed the weights eagerly, so the captured launches hold no pack jobs when the plan was fresh: a replay
        # must then make sure it still is (training between two sweeps, load_state_dict, ...)
        entry = (graph, s_eeg, s_spec, out, slots, plan, used)
        self._graphs[(tuple(eeg.shape), tuple(spec.shape))] = entry
        return entry

    def __call__(self, eeg, spec):
        entry = self._graphs.get((tuple(eeg.shape), tuple(spec.shape)))
        if entry is None:
            if eeg.shape[0] != spec.shape[0] or eeg.shape[0] == 0:
                raise RuntimeError(f"GradCamSweep: bad batch {tuple(eeg.shape)} / {tuple(spec.shape)}")
            entry = self._capture(eeg, spec)
        graph, s_eeg, s_spec, out, slots, plan, used = entry
        if plan is not None and not plan.fresh():
            if getattr(self.model.spectrogram_model, "_pack_plan", None) is not plan:
                # the parameters moved to other storage (a new FlatAdamW arena): the graph's operand buffers are orphaned
                self._graphs.clear()
                return self.__call__(eeg, spec)
            plan.run()                                   # repack eagerly; the graph's convolutions read the same operand buffers
        ptrs = []
        for t, st, via_slot in ((eeg, s_eeg, used[0]), (spec, s_spec, used[1])):
            if via_slot and self._slot_ready(t):
                ptrs.append(t.data_ptr())
            else:
                st.copy_(t, non_blocking=True)
                ptrs.append(st.data_ptr())
        if slots is not None:
            L.check(L.load().bx_store_u64x2(slots.data_ptr(), ptrs[0], ptrs[1], ops._stream()), "bx_store_u64x2")
        graph.replay()
        return out


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of ``n`` samples for ``rank`` of ``world``: sizes differ by at most one, lower ranks take the
    remainder, every sample belongs to exactly one rank (attribution sweeps shard by sample, no collective)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_sweep(fn, n_samples, batch_size, fetch, rank=None, world=None, gather=True, group=None):
    """Run an attribution ``fn`` over ``n_samples`` samples sharded contiguously over the ranks of the process group
    (BASELINE configs[3] / configs[4]).  ``fetch(lo, hi)`` returns the inputs (a tuple of tensors) of samples [lo, hi) already on
    this rank's device; ``fn(*inputs)`` returns a tensor whose first axis is the sample axis (e.g. Grad-CAM maps [b, 6, H, W]
    from a GradCamSweep, or integrated-gradients attributions).  Each rank walks its shard in batches of ``batch_size`` (the last
    one ragged).  ``gather=True``: rank 0 returns the [n_samples, ...] result in sample order (one padded ``dist.gather``), the
    other ranks None; ``gather=False``: every rank returns (lo, its shard's result).  Without a process group: one shard."""
    import torch.distributed as dist
    have_pg = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank(group) if have_pg else 0
    if world is None:
        world = dist.get_world_size(group) if have_pg else 1
    lo, hi = shard_bounds(n_samples, rank, world)
    if gather and world > 1 and n_samples < world:
        # decided from (n_samples, world), which every rank knows, BEFORE any collective: all ranks raise together instead of the
        # empty ones raising while the others wait in dist.gather
        raise RuntimeError("sharded_sweep(gather=True) needs at least one sample per rank; use gather=False for tiny sweeps")
    parts = []
    for b0 in range(lo, hi, batch_size):
        b1 = min(hi, b0 + batch_size)
        parts.append(fn(*fetch(b0, b1)).clone())         # clone: fn may return a static graph buffer
    mine = torch.cat(parts) if parts else None
    if not gather:
        return lo, mine
    if world == 1:
        return mine
    # every rank needs the trailing shape to build its padded block: take it from a rank that has samples
    cap = shard_bounds(n_samples, 0, world)[1]           # rank 0 holds the largest shard
    pad = torch.zeros(cap, *mine.shape[1:], dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    blocks = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, blocks, dst=0, group=group)
    if rank != 0:
        return None
    sizes = [shard_bounds(n_samples, r, world) for r in range(world)]
    return torch.cat([blk[:b - a] for blk, (a, b) in zip(blocks, sizes)])


def grad_cam(model, eeg, spec, target_layer="spectrogram_model.block5", class_idx=None, upsample=True, relu=True,
             return_parts=False):
    """Grad-CAM heat-maps of ``model(eeg, spec)``.

    score y_c = the model's output log-probability of class c;  w[b,k] = mean_hw dy_c/dA[b,k];
    cam[b] = ReLU(sum_k w[b,k] A[b,k]); optionally bilinear-upsampled to the spectrogram's H x W.

    target_layer: 'spectrogram_model.blockN' (stage output) or 'spectrogram_model.blockN.convK'
                  (that convolution's pre-ReLU output, i.e. what a hook on the reference's nn.Conv2d sees).
    class_idx:    None -> each sample's arg-max class; int -> that class; 'all' -> every class, output
                  gains a class axis [B, n_classes, H, W].
    """
    m = _TARGET.match(target_layer)
    if not m:
        raise ValueError(f"unsupported Grad-CAM target {target_layer!r}; use 'spectrogram_model.blockN[.convK]'")
    spec_model = model.spectrogram_model if hasattr(model, "spectrogram_model") else model
    blk = getattr(spec_model, f"block{m.group(1)}")
    conv_k = int(m.group(2)) if m.group(2) else 0
    if conv_k == 0 and m.group(1) == "5" and hasattr(model, "eeg_model") and hasattr(model, "fc1"):
        was_training = model.training
        model.eval()
        try:
            return _grad_cam_last_stage(model, eeg, spec, class_idx, upsample, relu, return_parts)
        finally:
            model.train(was_training)
    with _eval_frozen(model):
        spec_in = spec.detach().clone().requires_grad_(True)
        grabbed = {}
        hooks = []
        if conv_k:
            blk._preact, blk._capture = conv_k, {}
            hooks.append(blk.register_forward_pre_hook(lambda _m, args: grabbed.__setitem__("x", args[0])))
        else:
            hooks.append(blk.register_forward_hook(lambda _m, _i, o: grabbed.__setitem__("A", o)))
        try:
            out = model(eeg, spec_in) if hasattr(model, "spectrogram_model") else model(spec_in)
        finally:
            for h in hooks:
                h.remove()
        n_cls = out.shape[1]
        B = out.shape[0]
        if class_idx is None:
            seeds = [_class_seed(out, -1)]
        elif isinstance(class_idx, str):
            if class_idx != "all":
                raise ValueError(class_idx)
            seeds = [_class_seed(out, c) for c in range(n_cls)]
        else:
            seeds = [_class_seed(out, int(class_idx))]
        try:
            grads = []
            for sd in seeds:
                if conv_k:
                    torch.autograd.grad(out, grabbed["x"], grad_outputs=sd, retain_graph=True)
                    grads.append(blk._capture["grad"])
                else:
                    (g,) = torch.autograd.grad(out, grabbed["A"], grad_outputs=sd, retain_graph=True)
                    grads.append(g.permute(0, 2, 3, 1).contiguous())
            A = blk._capture["act"] if conv_k else grabbed["A"].detach().permute(0, 2, 3, 1).contiguous()
        finally:
            blk._preact, blk._capture = 0, None
        nm = len(grads)
        # map index = sample * nm + class
        G = torch.stack(grads, dim=1).reshape(B * nm, *grads[0].shape[1:]) if nm > 1 else grads[0]
        cam, wts = _reduce(A, G, nm, relu=relu)
        raw = _reduce(A, G, nm, relu=False)[0] if (return_parts and relu) else cam
        if upsample:
            cam = resize_bilinear(cam, spec.shape[-2:])
    stacked = isinstance(class_idx, str)
    def shape(t):
        return t.reshape(B, nm, *t.shape[1:]) if stacked else t
    if return_parts:
        return shape(cam), shape(raw), shape(wts), A, out.detach()
    return shape(cam)


def _abs(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(t)
    L.check(L.load().bx_abs(_p(t), _p(out), t.numel(), _stream()), "bx_abs")
    return out


def saliency(model, eeg, spec, reference_quirk=False):
    """|d max-logprob / d input| (reference XAI_Multimodality.py:3109-3129): returns
    (eeg_sal [B,Chans,T], spec_sal [B,H,W]); the spectrogram map is the max over channels.
    ``reference_quirk=True`` doubles the spectrogram map as the reference's second backward() does."""
    with _eval_frozen(model):
        e = eeg.detach().clone().float().requires_grad_(True)
        s = spec.detach().clone().float().requires_grad_(True)
        out = model(e, s)
        seed = _class_seed(out, -1)
        ge, gs = torch.autograd.grad(out, (e, s), grad_outputs=seed)
    B, Cc, H, W = gs.shape
    g_nhwc = ops.to_nhwc(gs, torch.float32)
    smap = torch.empty(B, H, W, dtype=torch.float32, device=gs.device)
    L.check(L.load().bx_saliency_reduce(_p(g_nhwc), _p(smap), B, H * W, Cc, g_nhwc.shape[3], 2.0 if reference_quirk else 1.0,
                                        L.BX_F32, _stream()), "bx_saliency_reduce")
    return _abs(ge.contiguous())[:, 0], smap


def generate_saliency_maps(model, dataloader, plot_eeg=None, plot_spectrogram=None, device=None):
    """Reference signature (XAI_Multimodality.py:3101).  Iterates ``((eeg, spec), label)`` batches, computes the
    reference's maps (batch element 0 of each batch, spectrogram map with its 2x accumulation quirk) and hands
    them to the optional plot callbacks; also returns them as a list of (eeg_map, spec_map) numpy pairs."""
    device = device or next(model.parameters()).device
    results = []
    for (eeg_data, spectrogram_data), _label in dataloader:
        e, s = eeg_data.to(device)[:1], spectrogram_data.to(device)[:1]
        es, ss = saliency(model, e, s, reference_quirk=True)
        pair = (es[0].cpu().numpy(), ss[0].cpu().numpy())
        results.append(pair)
        if plot_eeg is not None:
            plot_eeg(pair[0])
        if plot_spectrogram is not None:
            plot_spectrogram(pair[1])
    return results


def ig_nodes(n_steps=50):
    x, w = np.polynomial.legendre.leggauss(n_steps)
    return 0.5 * (1.0 + x), 0.5 * w


def _axpby(x, y, alpha, beta):
    L.check(L.load().bx_axpby(_p(x), _p(y), x.numel(), float(alpha), float(beta), _stream()), "bx_axpby")


def integrated_gradients(model, inputs, baselines=None, target=None, n_steps=50, max_batch=1024):
    """(x - x') * sum_k w_k grad F_target(x' + a_k (x - x')) with Gauss-Legendre nodes (Captum's default rule).
    inputs = (eeg [B,1,Ch,T], spec [B,C,H,W]); the k-loop is batched: up to ``max_batch`` interpolants per pass (sized for
    288 GB of HBM: a pass keeps ~18 MB of activations per interpolant at the benchmark shapes in bf16 storage, 35 MB in fp32;
    measured at 50 x B=64: 256 -> 1380, 512 -> 1415, 1024 -> 1472 samples/s -- the late stages stop being latency-bound)."""
    eeg, spec = (t.detach().float().contiguous() for t in inputs)
    be, bs = baselines if baselines is not None else (torch.zeros_like(eeg), torch.zeros_like(spec))
    be, bs = be.float().contiguous(), bs.float().contiguous()
    B = eeg.shape[0]
    alphas, steps = ig_nodes(n_steps)
    acc_e, acc_s = torch.zeros_like(eeg), torch.zeros_like(spec)
    alphas_dev = torch.tensor([float(a) for a in alphas], dtype=torch.float32, device=eeg.device)
    steps_dev = torch.tensor([float(w) for w in steps], dtype=torch.float32, device=eeg.device)
    per_pass = max(1, max_batch // B)
    # the kernels address an activation tensor with 32-bit byte offsets: keep the largest one of a pass (stage 1: H x W x 16
    # channels) under 2 GiB
    sm_ = getattr(model, "spectrogram_model", None)
    esize = 2 if getattr(sm_, "compute_dtype", torch.float32) == torch.bfloat16 else 4
    cap = ((1 << 31) - 1) // max(1, spec.shape[2] * spec.shape[3] * 16 * esize)
    per_pass = max(1, min(per_pass, cap // B))
    with _eval_frozen(model):
        with torch.no_grad():
            base_out = model(eeg, spec)                     # arg-max class of the un-interpolated input (Captum: target of the input)
        tmode = -1 if target is None else int(target)
        for k0 in range(0, n_steps, per_pass):
            ks = range(k0, min(n_steps, k0 + per_pass))
            xe = torch.empty(len(ks), *eeg.shape, dtype=torch.float32, device=eeg.device)
            xs = torch.empty(len(ks), *spec.shape, dtype=torch.float32, device=spec.device)
            a_dev, w_dev = alphas_dev[k0:k0 + len(ks)], steps_dev[k0:k0 + len(ks)]
            lib_ = L.load()
            L.check(lib_.bx_ig_interpolate(_p(eeg), _p(be), _p(a_dev), _p(xe), eeg.numel(), len(ks), _stream()), "bx_ig_interpolate")
            L.check(lib_.bx_ig_interpolate(_p(spec), _p(bs), _p(a_dev), _p(xs), spec.numel(), len(ks), _stream()), "bx_ig_interpolate")
            xe = xe.flatten(0, 1).requires_grad_(True)
            xs = xs.flatten(0, 1).requires_grad_(True)
            out = model(xe, xs)
            seed = _class_seed(base_out, tmode, rows=len(ks) * B)      # row k*B + b -> class of sample b
            ge, gs = torch.autograd.grad(out, (xe, xs), grad_outputs=seed)
            ge, gs = ge.contiguous(), gs.contiguous()
            L.check(lib_.bx_ig_accumulate(_p(ge), _p(w_dev), _p(acc_e), acc_e.numel(), len(ks), _stream()), "bx_ig_accumulate")
            L.check(lib_.bx_ig_accumulate(_p(gs), _p(w_dev), _p(acc_s), acc_s.numel(), len(ks), _stream()), "bx_ig_accumulate")
    de, ds = eeg.clone(), spec.clone()
    _axpby(be, de, -1.0, 1.0)
    _axpby(bs, ds, -1.0, 1.0)
    lib = L.load()
    L.check(lib.bx_mul(_p(acc_e), _p(de), _p(acc_e), acc_e.numel(), _stream()), "bx_mul")
    L.check(lib.bx_mul(_p(acc_s), _p(ds), _p(acc_s), acc_s.numel(), _stream()), "bx_mul")
    return acc_e, acc_s


def expected_gradients(model, x, background, nsamples=200, seed=0, max_batch=256):
    """SHAP GradientExplainer's estimator for a single-input model (the reference explains ``multimodal_model.eeg_model``,
    XAI_Multimodality.py:2283-2290):  phi_c(x) = E_{b, a}[(x - b) * d f_c/dx (b + a (x - b))], b drawn from ``background``,
    a ~ U(0,1); draws from numpy's default_rng(seed) in sample-major order.  Returns [B, n_classes, *x.shape[1:]].
    Interpolants are built with bx_axpby, gradients come from the HIP backward, products/means from bx_mul / bx_axpby."""
    x = x.detach().float().contiguous()
    background = background.detach().float().contiguous()
    rng = np.random.default_rng(seed)
    B = x.shape[0]
    lib = L.load()
    with _eval_frozen(model):
        with torch.no_grad():
            n_cls = model(x[:1]).shape[1]
        out = torch.zeros(B, n_cls, *x.shape[1:], dtype=torch.float32, device=x.device)
        for i in range(B):
            idx = rng.integers(0, background.shape[0], size=nsamples)
            alpha = rng.random(nsamples).astype(np.float32)
            for k0 in range(0, nsamples, max_batch):
                ks = range(k0, min(nsamples, k0 + max_batch))
                base = background[torch.as_tensor(idx[k0:k0 + len(ks)], device=x.device)].contiguous()
                diff = x[i:i + 1].expand_as(base).contiguous()
                _axpby(base, diff, -1.0, 1.0)                               # diff = x - b
                xi = base.clone()
                for j, k in enumerate(ks):
                    _axpby(diff[j], xi[j], float(alpha[k]), 1.0)            # b + a (x - b)
                xi.requires_grad_(True)
                y = model(xi)
                for c in range(n_cls):
                    seed_c = _class_seed(y, c)
                    (g,) = torch.autograd.grad(y, xi, grad_outputs=seed_c, retain_graph=True)
                    g = g.contiguous()
                    L.check(lib.bx_mul(_p(g), _p(diff), _p(g), g.numel(), _stream()), "bx_mul")
                    for j in range(len(ks)):
                        _axpby(g[j], out[i, c], 1.0 / nsamples, 1.0)
    return out


def predict_fn(images, model, device=None, max_batch=256):
    """LIME's batched-inference callback (reference XAI_Multimodality.py:1567-1574, :2710-2717; called by
    ``lime_image.LimeImageExplainer.explain_instance`` with the perturbed copies of one spectrogram image).

    images: sequence / array of H x W x C images (LIME passes float arrays holding 0..255 values; the reference casts them with
    ``astype(np.uint8)`` and applies torchvision's ToTensor = x / 255, channels first).  model: a single-input spectrogram model
    (``Spectrogram_Model`` or ``multimodal.forward_spectrogram``).  Returns ``softmax(model(batch))`` as a numpy array [N, classes]
    exactly like the reference (the model already ends in LogSoftmax, so these are its class probabilities).
    The uint8 -> channels-last conversion, the forward pass and the softmax all run on the GPU, ``max_batch`` images per pass."""
    imgs = np.ascontiguousarray(np.stack([np.asarray(im) for im in images]).astype(np.uint8))
    if imgs.ndim != 4:
        raise ValueError(f"predict_fn expects images [N, H, W, C], got {imgs.shape}")
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("brainxai.predict_fn: the model must live on the GPU; there is no CPU path")
    fwd = model.forward_spectrogram if hasattr(model, "forward_spectrogram") else model
    net = model.spectrogram_model if hasattr(model, "spectrogram_model") else model
    dt = getattr(net, "compute_dtype", torch.float32)
    was_training = model.training
    model.eval()
    lib = L.load()
    N, H, W, Cc = imgs.shape
    out = []
    try:
        with torch.no_grad():
            for i0 in range(0, N, max_batch):
                chunk = torch.from_numpy(imgs[i0:i0 + max_batch]).to(device)
                n = chunk.shape[0]
                x = torch.empty(n, H, W, ops.pad8(Cc), dtype=dt, device=device)
                L.check(lib.bx_u8_to_nhwc(_p(chunk), _p(x), n, H, W, Cc, ops.pad8(Cc), 1.0 / 255.0, ops.bx_dtype(dt), _stream()), "bx_u8_to_nhwc")
                logp = fwd(x.permute(0, 3, 1, 2)).float().contiguous()       # a logical-NCHW view of the internal layout: no further copy
                probs = torch.empty_like(logp)
                L.check(lib.bx_softmax_rows(_p(logp), _p(probs), n, logp.shape[1], _stream()), "bx_softmax_rows")
                out.append(probs.cpu())
    finally:
        model.train(was_training)
    return torch.cat(out).numpy()
