"""Training loops, the fused flat-arena AdamW and the data-parallel wrapper.

Reference anchors: train_and_validate_combined = XAI_Multimodality.py:1579-1681 (AdamW lr 1e-3, NB:1988;
KLDivLoss, NB:1989); load/save_checkpoint = XAI_Multimodality.py:279-313; setup / cleanup /
create_ddp_model = XAI_Multimodality.py:66-80; train_and_validate_eeg_distributed =
root/src/training/training_distributed.py:22-141 (DDP semantics: gradients averaged over ranks, BatchNorm
statistics local to a rank, parameters/buffers broadcast from rank 0, manual L2 penalty sum(p^2)*weight_decay).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import _p, _stream


# ------------------------------------------------------------------------------------------------
class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics over ONE flat fp32 parameter arena and ONE flat gradient arena.

    Construction re-points every parameter's ``.data`` at a slice of the arena (values preserved) and
    registers matching slices of the gradient arena with the autograd Functions, so weight-gradient
    kernels write straight into it.  ``step`` is a single fused kernel launch; the data-parallel
    wrapper all-reduces the gradient arena in place.

    It IS a ``torch.optim.Optimizer`` (one param group), so ``torch.optim.lr_scheduler`` classes drive it like any other
    optimizer (the reference's DDP loop steps a ReduceLROnPlateau, training_distributed.py:98-101).  The hyper-parameters
    live in a small device buffer that ``step`` refreshes only when ``param_groups[0]`` changed, so a step replayed from a
    hipGraph follows the schedule too.  ``l2_lambda`` > 0 fuses the reference's manual L2 penalty
    (training_distributed.py:52-57) into the same launch: the gradient 2*lambda*p is added on the fly and
    ``l2_value`` (device scalar) receives lambda * sum p^2 of the weights the step started from.

    Gradient accumulation over several backward passes is not supported: the weight-gradient kernels WRITE their arena
    slice (``backward()`` twice without ``zero_grad()`` raises instead of silently doubling the last gradient).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, allow_host=False):
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = plist[0].device
        if dev.type != "cuda" and not allow_host:
            # no silent CPU fallback: host tensors are only accepted when a test of the data-parallel plumbing
            # (gloo, no GPU) asks for it explicitly
            raise RuntimeError("brainxai.FlatAdamW: parameters must live on the GPU (allow_host=True is for the gloo logic tests only)")
        if any(p.device != dev or p.dtype != torch.float32 for p in plist):
            raise ValueError("FlatAdamW: parameters must be float32 on one device")
        super().__init__(plist, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW: one parameter group only")
        self.params = plist
        self.n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                view = self.flat_p.narrow(0, off, p.numel()).view(p.shape)
                view.copy_(p.data)
                p.data = view
                self.offsets.append(off)
                off += p.numel()
        self._grad_key = ops.register_grad_views(self.params, self.flat_g)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        # [step count | ticket counters of bx_adamw_step_dev]
        self._step_buf = torch.zeros(int(L.load().bx_adamw_step_words(self.n)) if dev.type == "cuda" else 1, dtype=torch.float32, device=dev)
        self.step_t = self._step_buf[:1]
        self.grad_scale = 1.0
        self.l2_lambda = 0.0
        self.is_cuda = dev.type == "cuda"
        if self.is_cuda:
            self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
            self._hyper_seen = None
            self.l2_value = torch.zeros((), dtype=torch.float32, device=dev)
            self._l2_partials = torch.zeros(int(L.load().bx_adamw_partials(self.n)), dtype=torch.float32, device=dev)
            self.refresh_hyper()

    def close(self):
        """Forget the gradient-arena registration (otherwise done when the optimizer is garbage-collected)."""
        ops.unregister_grad_views(getattr(self, "_grad_key", None))
        self._grad_key = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    def refresh_hyper(self):
        """Push param_groups[0] / grad_scale / l2_lambda to the device buffer when they changed since the last call."""
        g = self.param_groups[0]
        vals = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(self.grad_scale), float(self.l2_lambda), 0.0)
        if vals != self._hyper_seen:
            # the values ride as kernel arguments of a one-wave store, ordered on the current stream: no host staging buffer
            # whose reuse could race with a copy still queued behind many replayed steps (a per-step scheduler under hipGraph
            # replay runs the host far ahead of the GPU)
            import ctypes
            host = (ctypes.c_float * 8)(*vals)
            L.check(L.load().bx_store_f32x8(_p(self.hyper), ctypes.addressof(host), _stream()), "bx_store_f32x8")
            self._hyper_seen = vals

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.zero_()

    def gather_grads(self, first=0, last=None):
        """Make the gradient arena hold the gradients of parameters [first, last) (default: all); a no-op for slices the
        kernels wrote, zero for parameters that received no gradient."""
        if self.is_cuda:
            ops.join_side_streams(self.flat_g.device)      # weight-gradient kernels may still run on the side stream
            ops.wgrad_flush(self.flat_g.device)            # normally done by backward's end callback; no-op then
        base = self.flat_g.data_ptr()
        last = len(self.params) if last is None else last
        for p, off in zip(self.params[first:last], self.offsets[first:last]):
            slot = self.flat_g.narrow(0, off, p.numel())
            if p.grad is None:
                slot.zero_()
            elif p.grad.data_ptr() != base + 4 * off:
                slot.copy_(p.grad.reshape(-1))

    @torch.no_grad()
    def step(self, closure=None, gathered=False):
        if closure is not None:
            raise RuntimeError("FlatAdamW.step: closures are not supported")
        if not gathered:
            self.gather_grads()
        g = self.param_groups[0]
        ops.bump_param_epoch()                             # the kernel rewrites the arena behind torch's version counters (PackPlan.fresh)
        if self.is_cuda:
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing:
                self.refresh_hyper()                       # (inside a capture nothing can have changed since the eager step before it)
            want_l2 = self.l2_lambda > 0
            L.check(L.load().bx_adamw_step_dev(_p(self.flat_p), _p(self.flat_g), _p(self.exp_avg), _p(self.exp_avg_sq), self.n,
                                               _p(self.hyper), _p(self.step_t), _p(self._l2_partials) if want_l2 else None,
                                               _p(self.l2_value) if want_l2 else None, _stream()), "bx_adamw_step_dev")
        else:  # host tensors: only reached by the gloo/CPU logic tests of the data-parallel wrapper
            self.step_t += 1
            t = float(self.step_t)
            b1, b2 = g["betas"]
            gr = self.flat_g * self.grad_scale + 2.0 * self.l2_lambda * self.flat_p
            self.flat_p.mul_(1 - g["lr"] * g["weight_decay"])
            self.exp_avg.mul_(b1).add_(gr, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(gr, gr, value=1 - b2)
            denom = (self.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(g["eps"])
            self.flat_p.addcdiv_(self.exp_avg, denom, value=-g["lr"] / (1 - b1 ** t))

    def state_dict(self):
        return {"flat_adamw": 1, "step": self.step_t.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if "flat_adamw" not in sd:
            raise ValueError("FlatAdamW.load_state_dict: not a FlatAdamW state")
        self.step_t.copy_(sd["step"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v


# ------------------------------------------------------------------------------------------------
def setup(rank, world_size, backend=None):
    """dist.init_process_group (reference XAI_Multimodality.py:66-68); backend 'nccl' is RCCL on ROCm."""
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world_size)


# What this library itself created against the live process group: asynchronous-reduction handles (each holds a c10d Work, i.e.
# RCCL stream events and a reference to the communicator) and graphed steps (hipGraphs captured while the group was alive; the
# one-graph data-parallel step even contains the collectives, and RCCL requires a communicator to outlive every graph that
# captured one of its operations).  cleanup() releases them before the communicator goes -- the reference's calling pattern is a
# bare ``cleanup()`` with model, wrapper and step objects all still alive (XAI_Multimodality.py:70-71).
import weakref as _weakref

_LIVE_REDUCTIONS = _weakref.WeakSet()
_LIVE_STEPPERS = _weakref.WeakSet()


def release_group_objects():
    """Wait for and drop every outstanding reduction handle, drop every captured graph of a data-parallel GraphedTrainStep (they
    are re-captured on their next use).  Returns (handles released, graphs released)."""
    nred = ngraph = 0
    for r in list(_LIVE_REDUCTIONS):
        nred += 1 if r.finish() else 0
    for st in list(_LIVE_STEPPERS):
        ngraph += st.release()
    return nred, ngraph


def cleanup():
    """dist.destroy_process_group() (reference XAI_Multimodality.py:70-71), made safe for what brainxai keeps alive: outstanding
    reductions are waited for and their c10d Work handles dropped, graphs captured under the group are destroyed, the device is
    drained -- then the communicator is torn down."""
    if dist.is_initialized():
        release_group_objects()
        if torch.cuda.is_available():
            torch.cuda.synchronize()                      # nothing may still be queued on RCCL's stream when the communicator goes
        dist.destroy_process_group()


class DataParallel(nn.Module):
    """One-process-per-GPU data parallelism with the reference's DDP semantics.

    * construction broadcasts parameters and buffers from rank 0 (DDP constructor);
    * ``sync_gradients()`` (call between backward and step) averages gradients over ranks with ONE
      all-reduce of the flat gradient arena (FlatAdamW) -- RCCL over xGMI on GPUs, gloo in CPU tests;
    * BatchNorm statistics stay local to each rank (no SyncBN), buffers are NOT re-broadcast per step.
    ``state_dict()`` prefixes keys with ``module.`` exactly like torch's wrapper.
    """

    def __init__(self, module, device_ids=None, output_device=None, process_group=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._flat_fallback = None
        if self.world_size > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=self.group)

    def forward(self, *args, **kw):
        return self.module(*args, **kw)

    def reduce_async(self, flat_slice):
        """Start the mean all-reduce of one contiguous slice of a gradient arena (RCCL over xGMI on GPUs; it runs on the
        process group's own stream, ordered behind what the current stream has queued so far).  ``wait()`` on the result makes
        the current stream wait for it.  None without an initialised process group."""
        if not dist.is_initialized():
            return None
        return _Reduction(flat_slice, self.group, self.world_size)

    def sync_gradients(self, optimizer=None):
        if not dist.is_initialized():
            return          # plain single-process use; with an initialised group the collective runs even for world_size 1
        if isinstance(optimizer, FlatAdamW):
            optimizer.gather_grads()
            flat = optimizer.flat_g
        else:
            if torch.cuda.is_available():
                ops.join_side_streams()
            grads = [p.grad for p in self.module.parameters() if p.grad is not None]
            flat = torch.cat([g.reshape(-1) for g in grads])
        if flat.is_cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world_size)
        if not isinstance(optimizer, FlatAdamW):
            off = 0
            for g in grads:
                g.copy_(flat.narrow(0, off, g.numel()).view_as(g))
                off += g.numel()


class _Cut:
    """``cut`` callback of Spectrogram_Model.features: remembers the activation and continues from a detached leaf, so that
    ``loss.backward()`` stops there (late stages + heads + EEG branch) and ``finish()`` runs the rest (early stages)."""

    def __call__(self, x):
        self.x = x
        self.leaf = x.detach().requires_grad_(True)
        return self.leaf

    def finish(self):
        torch.autograd.backward([self.x], [self.leaf.grad])


class _Reduction:
    """Handle of an asynchronous mean all-reduce of one slice of the gradient arena."""

    def __init__(self, tensor, group, world):
        self.tensor, self.world = tensor, world
        self.div = not tensor.is_cuda                       # gloo has no AVG
        self.work = dist.all_reduce(tensor, op=dist.ReduceOp.SUM if self.div else dist.ReduceOp.AVG, group=group, async_op=True)
        _LIVE_REDUCTIONS.add(self)

    def wait(self):
        if self.work is None:
            return
        self.work.wait()                                    # NCCL/RCCL: the current stream waits for the collective's stream
        self.work = None                                    # the c10d Work (its events, its reference to the communicator) goes now
        _LIVE_REDUCTIONS.discard(self)
        if self.div:
            self.tensor.div_(self.world)

    def finish(self):
        """cleanup(): a handle nobody waited for is waited for now.  True when there was something to release."""
        if self.work is None:
            return False
        self.wait()
        return True


def overlap_plan(model, optimizer):
    """Where to cut the step of a MultimodalModel for the overlapped data-parallel step: (stage index k, parameter index of the
    first parameter after the cut, arena offset there), or None when the model / optimizer do not support it.  The arena follows
    ``model.parameters()``: [EEG branch | block1 | block2 | block3 .. block5, fc, fc1, fc2]; the cut after stage 2 leaves 88 % of
    the bytes (everything from block3 on) in the first bucket, which is complete when backward reaches block2."""
    sm = getattr(model, "spectrogram_model", None)
    if not isinstance(optimizer, FlatAdamW) or sm is None or not hasattr(sm, "block3") or not hasattr(model, "_fusable") or not model._fusable():
        return None
    first = sm.block3.conv1.weight
    for i, p in enumerate(optimizer.params):
        if p is first:
            return 2, i, optimizer.offsets[i]
    return None


def create_ddp_model(model, device_ids=None, output_device=None):
    """reference XAI_Multimodality.py:77-80."""
    if device_ids:
        model = model.to(device_ids[0])
    return DataParallel(model, device_ids=device_ids, output_device=output_device)


# ------------------------------------------------------------------------------------------------
def train_step_overlapped(model, optimizer, eeg, spec, labels, criterion, ddp, plan=None):
    """The data-parallel step with the gradient exchange overlapped with backward (DDP's bucketing, done with two buckets of the
    flat arena): autograd is cut after spectrogram stage 2; ``loss.backward()`` yields the gradients of stages 3-5, the heads
    and the EEG branch's slice; their all-reduce (88 % of the bytes) then runs on RCCL's stream while stages 2-1 run
    their backward; the small second bucket follows; the fused AdamW waits for both.  Same arithmetic as train_step(ddp=...)."""
    plan = plan or overlap_plan(model, optimizer)
    if plan is None:
        return train_step(model, optimizer, eeg, spec, labels, criterion, ddp=ddp)
    k, pidx, off = plan
    cut = _Cut()
    optimizer.zero_grad()
    out = model(eeg, spec, cut=(k, cut))
    loss = criterion(out, labels)
    loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
    optimizer.gather_grads(pidx, None)
    r1 = ddp.reduce_async(optimizer.flat_g.narrow(0, off, optimizer.n - off))
    cut.finish()
    optimizer.gather_grads(0, pidx)
    r2 = ddp.reduce_async(optimizer.flat_g.narrow(0, 0, off))
    for r in (r1, r2):
        if r is not None:
            r.wait()
    optimizer.step(gathered=True)
    return loss.detach(), out.detach()


def train_step(model, optimizer, eeg, spec, labels, criterion, ddp=None):
    """zero_grad -> forward -> loss -> backward -> (all-reduce) -> step (reference NB:1597-1601).
    Returns (loss, outputs) as device tensors; nothing here synchronises with the host."""
    optimizer.zero_grad()
    out = model(eeg, spec)
    loss = criterion(out, labels)
    loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
    if ddp is not None:
        ddp.sync_gradients(optimizer)
        if isinstance(optimizer, FlatAdamW):
            optimizer.step(gathered=True)
            return loss.detach(), out.detach()
    optimizer.step()
    return loss.detach(), out.detach()


def load_checkpoint(checkpoint_dir, checkpoint_filename, model, optimizer):
    """reference XAI_Multimodality.py:279-304 (combined-loop variant returns 5 values, NB:1582)."""
    path = os.path.join(checkpoint_dir, checkpoint_filename)
    if os.path.isfile(path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"])
        optimizer.load_state_dict(ck["optimizer"])
        return ck["epoch"], ck["train_losses"], ck["valid_losses"], ck["train_accuracies"], ck["valid_accuracies"]
    return 0, [], [], [], []


def save_checkpoint(state, checkpoint_dir, checkpoint_filename):
    os.makedirs(checkpoint_dir, exist_ok=True)
    torch.save(state, os.path.join(checkpoint_dir, checkpoint_filename))


class GraphedTrainStep:
    """forward + loss + backward (+ fused AdamW when not data-parallel) of one training step replayed from a hipGraph.

    A step is ~100 kernel launches; issued one by one from Python the host becomes the bottleneck (2.44 ms vs 2.0 ms per
    step on MI355X).  The first batch of a given shape runs eagerly, the second is captured, later ones are replayed:
    inputs are copied into the graph's static buffers, the dropout seeds and BatchNorm statistics live on the device and
    advance inside the graph, and the optimizer's hyper-parameters are read from FlatAdamW's device buffer (refreshed before
    every replay when ``param_groups`` changed), so a learning-rate schedule is honoured without re-capturing.  Needs FlatAdamW
    (stable gradient arena) and CUDA inputs; anything else, or BX_GRAPH_LOOPS=0, falls back to the eager step.  Returned loss /
    output are the graph's static tensors (valid until the next call).

    (Round 1 carried an experimental six-graph "branch-parallel" replay, BX_BRANCH_GRAPHS: it left the eager trajectory at
    B=64 -- cross-queue visibility at graph launches -- gained 1.4 % and was removed in round 2)"""

    def __init__(self, model, optimizer, criterion, ddp=None, adopt_inputs=False, strict=False):
        """``adopt_inputs``: the tensors of the capturing call become the graph's static inputs themselves (no clone, no
        per-step copy when the same tensors are passed again) -- for callers that refill one resident batch in place.
        ``strict``: a failed capture raises instead of continuing eagerly (bench.py: a slower number must not pass silently)."""
        self.model, self.opt, self.crit, self.ddp, self.adopt, self.strict = model, optimizer, criterion, ddp, adopt_inputs, strict
        if ddp is not None:
            _LIVE_STEPPERS.add(self)                        # cleanup() destroys this step's graphs before the communicator
        self.enabled = isinstance(optimizer, FlatAdamW) and optimizer.is_cuda and os.environ.get("BX_GRAPH_LOOPS", "1") != "0"
        self._seen, self._graphs = set(), {}
        # data-parallel: the step is cut after spectrogram stage 2 and the gradient arena is exchanged in two buckets, the first
        # (everything from stage 3 on, 88 % of the bytes) beside the early stages' backward (BX_DDP_OVERLAP=0: one collective after
        # the whole backward).  Default since round 3: ONE hipGraph holds forward, backward, both all-reduces (forked onto RCCL's
        # stream inside the capture) and the fused AdamW, so a data-parallel step is one replay like the single-GPU step;
        # BX_DDP_GRAPH=pieces keeps round 2's form (graph -> all-reduce -> graph -> all-reduce -> eager AdamW, four host-side pieces).
        self.plan = overlap_plan(model, optimizer) if (ddp is not None and os.environ.get("BX_DDP_OVERLAP", "1") != "0") else None
        self.one_graph = ddp is not None and os.environ.get("BX_DDP_GRAPH", "one") != "pieces"

    def _finish(self):
        if self.ddp is not None:
            self.ddp.sync_gradients(self.opt)
            self.opt.step(gathered=True)

    def _capture_overlapped(self, static_in, static_lab):
        k, pidx, off = self.plan
        cut = _Cut()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            self.opt.zero_grad()
            out = self.model(*static_in, cut=(k, cut))
            loss = self.crit(out, static_lab)
            loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
            self.opt.gather_grads(pidx, None)
        with torch.cuda.graph(g2, pool=g1.pool()):          # always replayed right after g1: one memory pool
            cut.finish()
            self.opt.gather_grads(0, pidx)
        return ("overlap", (g1, g2), static_in, static_lab, loss.detach(), out.detach(), cut)

    def _capture_ddp_one(self, static_in, static_lab):
        """The whole data-parallel step in one graph.  The collectives are captured: ProcessGroupNCCL launches them on RCCL's stream,
        which the capture forks from the step's stream at the call and joins again at ``wait()`` -- bucket 1 therefore sits on a
        branch beside the backward of stages 2-1, exactly the eager step's dependencies.  Same kernels, same order per stream: the
        trajectory is bit-identical to the eager overlapped step (tests/test_gpu_parity.py, 1-rank RCCL group)."""
        cut = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.opt.zero_grad()
            if self.plan is not None and len(static_in) == 2:
                k, pidx, off = self.plan
                cut = _Cut()
                out = self.model(*static_in, cut=(k, cut))
                loss = self.crit(out, static_lab)
                loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
                self.opt.gather_grads(pidx, None)
                r1 = self.ddp.reduce_async(self.opt.flat_g.narrow(0, off, self.opt.n - off))
                cut.finish()
                self.opt.gather_grads(0, pidx)
                r2 = self.ddp.reduce_async(self.opt.flat_g.narrow(0, 0, off))
                for r in (r1, r2):
                    if r is not None:
                        r.wait()
            else:
                out = self.model(*static_in)
                loss = self.crit(out, static_lab)
                loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
                self.ddp.sync_gradients(self.opt)
            self.opt.step(gathered=True)
        return ("ddp_one", g, static_in, static_lab, loss.detach(), out.detach(), cut)

    def _replay_overlapped(self, entry):
        _, (g1, g2), _, _, loss, out, _ = entry
        _, _, off = self.plan
        g1.replay()
        r1 = self.ddp.reduce_async(self.opt.flat_g.narrow(0, off, self.opt.n - off))
        g2.replay()
        r2 = self.ddp.reduce_async(self.opt.flat_g.narrow(0, 0, off))
        for r in (r1, r2):
            if r is not None:
                r.wait()
        self.opt.step(gathered=True)
        return loss, out

    def release(self):
        """Drop the captured graphs (and with them their memory pools and static tensors); the next call of a known shape
        captures again.  Returns the number of graphs dropped."""
        n = 0
        for entry in self._graphs.values():
            n += len(entry[1]) if isinstance(entry[1], tuple) else 1
        if n and torch.cuda.is_available():
            torch.cuda.synchronize()                        # a replay may still be running: the graph objects die only after it
        self._graphs.clear()
        return n

    def _eager(self, inputs, labels):
        self.opt.zero_grad()
        out = self.model(*inputs)
        loss = self.crit(out, labels)
        loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
        if self.ddp is not None:
            self._finish()
        else:
            self.opt.step()
        return loss.detach(), out.detach()

    def _eager_any(self, inputs, labels):
        if self.plan is not None and len(inputs) == 2:
            return train_step_overlapped(self.model, self.opt, inputs[0], inputs[1], labels, self.crit, self.ddp, self.plan)
        return self._eager(inputs, labels)

    def __call__(self, inputs, labels):
        if not self.enabled or not self.model.training:
            return self._eager_any(inputs, labels)
        key = (tuple((tuple(t.shape), t.dtype) for t in inputs), tuple(labels.shape), labels.dtype)
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen:                 # first batch of this shape: plain step (also sizes every workspace)
                self._seen.add(key)
                return self._eager_any(inputs, labels)
            static_in = [t.detach() if self.adopt else t.detach().clone() for t in inputs]
            static_lab = labels.detach() if self.adopt else labels.detach().clone()
            try:
                torch.cuda.synchronize()
                entry = None
                if self.one_graph and dist.is_initialized():
                    try:
                        entry = self._capture_ddp_one(static_in, static_lab)
                    except Exception as exc:          # noqa: BLE001  (a stack whose RCCL cannot be captured: fall back to the pieces)
                        if self.strict:
                            raise
                        print(f"[brainxai] one-graph capture of the data-parallel step failed ({type(exc).__name__}: {exc}); "
                              "using graph + collective pieces")
                        self.one_graph = False
                        torch.cuda.synchronize()
                if entry is not None:
                    pass
                elif self.plan is not None and len(inputs) == 2:
                    entry = self._capture_overlapped(static_in, static_lab)
                else:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self.opt.zero_grad()
                        out = self.model(*static_in)
                        loss = self.crit(out, static_lab)
                        loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
                        if self.ddp is None:
                            self.opt.step()
                    entry = ("one", graph, static_in, static_lab, loss.detach(), out.detach())
            except Exception as exc:                  # noqa: BLE001  (capture is an optimisation, never a requirement)
                if self.strict:
                    raise
                print(f"[brainxai] hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly")
                self.enabled = False
                torch.cuda.synchronize()
                return self._eager_any(inputs, labels)
            self._graphs[key] = entry
        static_in, static_lab = entry[2], entry[3]
        for dst, src in zip(static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if static_lab.data_ptr() != labels.data_ptr():
            static_lab.copy_(labels, non_blocking=True)
        self.opt.refresh_hyper()                      # lr / betas / weight decay changed by a scheduler since the last step?
        ops.bump_param_epoch()                        # the replayed optimizer step rewrites the parameters
        if entry[0] == "overlap":
            return self._replay_overlapped(entry)
        if entry[0] == "ddp_one":
            entry[1].replay()
            self.opt._opt_called = True
            return entry[4], entry[5]
        _, graph, _, _, loss, out = entry
        graph.replay()
        self._finish()
        self.opt._opt_called = True                   # the optimizer step ran inside the graph: torch's LR schedulers look for this flag
        return loss, out


class AsyncCheckpointer:
    """the per-epoch checkpoint without stalling the training stream.  ``save(state)`` snapshots every
    CUDA tensor of the (nested) state dict into pinned host buffers with non-blocking copies on a side stream, and a worker
    thread waits for that stream's event, then ``torch.save``s the reference-layout dict to ``<name>.tmp`` and renames it
    over the previous file (a reader never sees a half-written checkpoint).  The next epoch's kernels run meanwhile.
    ``wait()`` joins outstanding writes (call it before reading the file or exiting)."""

    def __init__(self, checkpoint_dir, checkpoint_filename):
        import threading
        self.path = os.path.join(checkpoint_dir, checkpoint_filename)
        os.makedirs(checkpoint_dir, exist_ok=True)
        self._thread, self._threading = None, threading
        self._stream = None
        self._pinned = {}
        self.error = None

    def _snapshot(self, obj, key):
        if torch.is_tensor(obj):
            if not obj.is_cuda:
                return obj.detach().clone()
            buf = self._pinned.get(key)
            if buf is None or buf.shape != obj.shape or buf.dtype != obj.dtype:
                buf = self._pinned[key] = torch.empty(obj.shape, dtype=obj.dtype, device="cpu", pin_memory=True)
            buf.copy_(obj.detach(), non_blocking=True)
            return buf
        if isinstance(obj, dict):
            return type(obj)((k, self._snapshot(v, f"{key}/{k}")) for k, v in obj.items())
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._snapshot(v, f"{key}/{i}") for i, v in enumerate(obj))
        return obj

    def _clone_device(self, obj):
        """Device-side copy of every CUDA tensor, queued on the CURRENT stream: what the training stream writes next (BatchNorm
        running statistics in the first forward, the whole parameter arena in the fused AdamW) can then no longer reach the
        snapshot -- state_dict entries are views of FlatAdamW's arena and of the live buffers, not copies."""
        if torch.is_tensor(obj):
            return obj.detach().clone() if obj.is_cuda else obj
        if isinstance(obj, dict):
            return type(obj)((k, self._clone_device(v)) for k, v in obj.items())
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._clone_device(v) for v in obj)
        return obj

    def save(self, state):
        self.wait()                                        # one write in flight; also frees the pinned buffers for reuse
        stable = None
        if torch.cuda.is_available():
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            stable = self._clone_device(state)             # ordered on the training stream, BEFORE the next step's kernels
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                snap = self._snapshot(stable, "")          # device clones -> pinned host buffers, beside the next epoch
                done = torch.cuda.Event()
                done.record()
        else:
            snap, done = self._snapshot(state, ""), None

        def work(stable=stable):
            try:
                if done is not None:
                    done.synchronize()
                del stable                                 # the clones were only read on the side stream, which has finished
                cpu = _to_plain(snap)
                torch.save(cpu, self.path + ".tmp")
                os.replace(self.path + ".tmp", self.path)
            except Exception as exc:                       # noqa: BLE001  (surfaced by wait())
                self.error = exc
        stable = None
        self._thread = self._threading.Thread(target=work, daemon=True)
        self._thread.start()

    def wait(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self.error is not None:
            err, self.error = self.error, None
            raise RuntimeError(f"asynchronous checkpoint write failed: {err}") from err


def _to_plain(obj):
    """pinned snapshot -> ordinary CPU tensors (so that the file loads without CUDA / pinned-memory support)."""
    if torch.is_tensor(obj):
        return obj.clone() if obj.is_pinned() else obj
    if isinstance(obj, dict):
        return type(obj)((k, _to_plain(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_plain(v) for v in obj)
    return obj


def _run_epoch(model, loader, criterion, device, optimizer=None, ddp=None, unpack=None, per_batch_sum=False, stepper=None):
    """One pass; loss*B (combined loop, NB:1603) or the plain per-batch loss (DDP loop, training_distributed.py:59,89:
    ``per_batch_sum``) and the correct counts accumulate ON DEVICE (the reference syncs three times per step).
    ``stepper``: the caller's GraphedTrainStep (its captured graphs then survive from epoch to epoch and die with the caller's
    loop); without one a training pass builds its own for this epoch."""
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    correct = torch.zeros((), dtype=torch.int64, device=device)
    total = 0
    for batch in loader:
        inputs, labels = unpack(batch) if unpack else batch
        inputs = [t.to(device, non_blocking=True) for t in (inputs if isinstance(inputs, (tuple, list)) else (inputs,))]
        labels = labels.to(device, non_blocking=True)
        if optimizer is not None:
            if stepper is None:
                stepper = GraphedTrainStep(model, optimizer, criterion, ddp)
            loss, out = stepper(inputs, labels)
        else:
            with torch.no_grad():
                out = model(*inputs)
                loss = criterion(out, labels)
        n = labels.shape[0]
        loss_sum += loss.detach() if per_batch_sum else loss.detach() * n
        correct += (out.detach().argmax(1) == labels.argmax(1)).sum()
        total += n
    total = max(total, 1)
    return float(loss_sum) / total, float(correct) / total * 100.0


def train_and_validate_combined(model, train_loader, valid_loader, epochs, optimizer, criterion, device, checkpoint_dir,
                                n=2, sample_spectrogram=None, async_checkpoint=False):
    """Reference XAI_Multimodality.py:1579-1681 minus its LIME tail (CPU skimage; out of scope).
    Batches are ``((eeg, spec), labels)``; returns (train_losses, valid_losses, train_accuracies, valid_accuracies)
    and writes ``combined_checkpoint.pth.tar`` with the reference's dict layout each epoch (``async_checkpoint=True``:
    through AsyncCheckpointer, the file of the last epoch is complete when the function returns)."""
    name = "combined_checkpoint.pth.tar"
    start, tr_l, va_l, tr_a, va_a = load_checkpoint(checkpoint_dir, name, model, optimizer)
    saver = AsyncCheckpointer(checkpoint_dir, name) if async_checkpoint else None
    # the loop OWNS the graphed step: model, optimizer, captured hipGraphs and their memory pools are released when it returns
    stepper = GraphedTrainStep(model, optimizer, criterion)
    for epoch in range(start, epochs):
        model.train()
        l, a = _run_epoch(model, train_loader, criterion, device, optimizer, stepper=stepper)
        tr_l.append(l); tr_a.append(a)
        model.eval()
        l, a = _run_epoch(model, valid_loader, criterion, device)
        va_l.append(l); va_a.append(a)
        print(f"Epoch {epoch + 1}/{epochs}:\n  Train Loss: {tr_l[-1]:.4f} Train Accuracy: {tr_a[-1]:.2f}%\n"
              f"  Valid Loss: {va_l[-1]:.4f} Valid Accuracy: {va_a[-1]:.2f}%")
        state = {"epoch": epoch + 1, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                 "train_losses": list(tr_l), "valid_losses": list(va_l), "train_accuracies": list(tr_a), "valid_accuracies": list(va_a)}
        if saver is not None:
            saver.save(state)
        else:
            save_checkpoint(state, checkpoint_dir, name)
    if saver is not None:
        saver.wait()
    return tr_l, va_l, tr_a, va_a


def load_checkpoint_distributed(checkpoint_dir, checkpoint_filename, model, optimizer):
    """What the reference's DDP loop expects of load_checkpoint (training_distributed.py:31 unpacks SEVEN values; the committed
    data_utils.py:259-283 returns six -- SURVEY fact 7): the five of the combined loop + the learning-rate history + the
    per-epoch regularisation losses, so that a resumed run continues both lists."""
    path = os.path.join(checkpoint_dir, checkpoint_filename)
    if os.path.isfile(path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"])
        optimizer.load_state_dict(ck["optimizer"])
        return (ck["epoch"], ck["train_losses"], ck["valid_losses"], ck["train_accuracies"], ck["valid_accuracies"],
                list(ck.get("lr_scheduler", [])), list(ck.get("regularization_losses", [])))
    return 0, [], [], [], [], [], []


def l2_penalty_(model_or_params, weight_decay, optimizer=None):
    """The reference's manual penalty (training_distributed.py:52-57): ``reg = weight_decay * sum_p sum(p**2)`` over
    ``model.parameters()`` joins the loss, i.e. its gradient ``2 * weight_decay * p`` joins every ``p.grad``.
    With a FlatAdamW the work is deferred into the fused optimizer launch (``optimizer.l2_lambda``; the value lands in
    ``optimizer.l2_value`` when ``step`` runs) and this returns None; otherwise it is done here, one bx_sumsq + bx_axpby per
    parameter, and the value is returned as a device scalar."""
    wd = float(weight_decay)
    if isinstance(optimizer, FlatAdamW) and optimizer.is_cuda:
        optimizer.l2_lambda = wd
        return None
    params = list(model_or_params.parameters()) if isinstance(model_or_params, nn.Module) else list(model_or_params)
    lib = L.load()
    dev = params[0].device
    total = torch.zeros((), dtype=torch.float32, device=dev)
    part = torch.empty((), dtype=torch.float32, device=dev)
    for p in params:
        flat = p.detach().reshape(-1)
        if not flat.is_cuda:
            raise RuntimeError("brainxai.l2_penalty_: parameters must live on the GPU; there is no CPU path")
        L.check(lib.bx_sumsq(_p(flat), flat.numel(), _p(part), _stream()), "bx_sumsq")
        L.check(lib.bx_axpby(_p(part), _p(total), 1, wd, 1.0, _stream()), "bx_axpby")                  # total += wd * sum p^2
        if p.grad is not None:
            g = p.grad.reshape(-1)
            L.check(lib.bx_axpby(_p(flat), _p(g), flat.numel(), 2.0 * wd, 1.0, _stream()), "bx_axpby")   # grad += 2 wd p
    return total


def train_and_validate_eeg_distributed(model, train_loader, valid_loader, epochs, optimizer, criterion, scheduler, device,
                                       checkpoint_dir, logger, rank, world_size):
    """Reference root/src/training/training_distributed.py:22-141 for a single-input model (EEGNet), same bookkeeping:

    * ``total_loss = criterion(out, labels) + model.weight_decay * sum(p**2)`` (:52-57) -- the penalty's value and gradient
      come from the fused optimizer launch (FlatAdamW) or from l2_penalty_();
    * train / regularisation / validation losses are the SUMS of the per-batch values divided by the number of SAMPLES
      (:59-60, :70-71, :95), accuracies arg-max based;
    * ``scheduler.step(valid_loss)`` (:99; ReduceLROnPlateau -- other schedulers are stepped without an argument) and the
      learning rate it leaves is appended to the ``lr_scheduler`` history (:101);
    * rank 0 writes ``eeg_checkpoint.pth.tar`` with the reference's keys, ``module.``-prefixed weights (:104-117); a resumed run
      restores both histories (load_checkpoint_distributed).
    Nothing in the batch loop synchronises with the host: the running sums live on the device."""
    setup(rank, world_size)
    model = model.to(device)
    wd = float(getattr(model, "weight_decay", 0.0))
    ddp = DataParallel(model, device_ids=[rank] if torch.cuda.is_available() else None)
    name = "eeg_checkpoint.pth.tar"
    start, tr_l, va_l, tr_a, va_a, lr_hist, reg_losses = load_checkpoint_distributed(checkpoint_dir, name, ddp, optimizer)
    flat = isinstance(optimizer, FlatAdamW)
    plateau = isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau)
    for epoch in range(start, epochs):
        if logger:
            logger.info(f"Starting Epoch {epoch + 1}/{epochs}")
        model.train()
        if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(epoch)           # :42
        loss_sum = torch.zeros((), device=device); reg_sum = torch.zeros((), device=device)
        correct = torch.zeros((), dtype=torch.int64, device=device); total = 0
        for data, labels in train_loader:
            data, labels = data.to(device), labels.to(device)
            optimizer.zero_grad()
            out = ddp(data)
            loss = criterion(out, labels)
            loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
            reg = l2_penalty_(model, wd, optimizer) if wd > 0 else None
            ddp.sync_gradients(optimizer)
            optimizer.step(gathered=True) if flat else optimizer.step()
            if wd > 0:
                reg_sum += optimizer.l2_value if reg is None else reg
            loss_sum += loss.detach()
            correct += (out.detach().argmax(1) == labels.argmax(1)).sum(); total += labels.shape[0]
        tr_l.append(float(loss_sum) / max(total, 1)); tr_a.append(float(correct) / max(total, 1) * 100.0)
        reg_losses.append(float(reg_sum) / max(total, 1))
        model.eval()
        l, a = _run_epoch(ddp, valid_loader, criterion, device, per_batch_sum=True)
        va_l.append(l); va_a.append(a)
        if scheduler is not None:
            scheduler.step(va_l[-1]) if plateau else scheduler.step()
            lr_hist.append(float(scheduler.get_last_lr()[0]) if hasattr(scheduler, "get_last_lr") else float(optimizer.param_groups[0]["lr"]))
        if logger:
            logger.info(f"Epoch {epoch + 1}/{epochs} train {tr_l[-1]:.5f} reg {reg_losses[-1]:.5f} / {tr_a[-1]:.2f}% valid {va_l[-1]:.5f}/{va_a[-1]:.2f}%")
        if rank == 0:
            save_checkpoint({"epoch": epoch + 1, "state_dict": ddp.state_dict(), "optimizer": optimizer.state_dict(),
                             "train_losses": tr_l, "valid_losses": va_l, "train_accuracies": tr_a, "valid_accuracies": va_a,
                             "lr_scheduler": lr_hist, "regularization_losses": reg_losses}, checkpoint_dir, name)
    return tr_l, va_l, tr_a, va_a
