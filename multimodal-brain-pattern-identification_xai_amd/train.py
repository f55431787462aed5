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
class FlatAdamW:
    """torch.optim.AdamW semantics over ONE flat fp32 parameter arena and ONE flat gradient arena.

    Construction re-points every parameter's ``.data`` at a slice of the arena (values preserved) and
    registers matching slices of the gradient arena with the autograd Functions, so weight-gradient
    kernels write straight into it.  ``step`` is a single fused kernel launch; the data-parallel
    wrapper all-reduces the gradient arena in place.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, allow_host=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda" and not allow_host:
            # no silent CPU fallback: host tensors are only accepted when a test of the data-parallel plumbing
            # (gloo, no GPU) asks for it explicitly
            raise RuntimeError("brainxai.FlatAdamW: parameters must live on the GPU (allow_host=True is for the gloo logic tests only)")
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdamW: parameters must be float32 on one device")
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
        ops.register_grad_views(self.params, self.flat_g)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.grad_scale = 1.0
        self.is_cuda = dev.type == "cuda"

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.zero_()

    def gather_grads(self):
        """Make the gradient arena hold every parameter's gradient (no-op for slices the kernels wrote)."""
        if self.is_cuda:
            ops.join_side_streams(self.flat_g.device)      # weight-gradient kernels may still run on the side stream
            ops.wgrad_flush(self.flat_g.device)            # normally done by backward's end callback; no-op then
        base = self.flat_g.data_ptr()
        for p, off in zip(self.params, self.offsets):
            slot = self.flat_g.narrow(0, off, p.numel())
            if p.grad is None:
                slot.zero_()
            elif p.grad.data_ptr() != base + 4 * off:
                slot.copy_(p.grad.reshape(-1))

    @torch.no_grad()
    def step(self, gathered=False):
        if not gathered:
            self.gather_grads()
        g = self.param_groups[0]
        if self.is_cuda:
            L.check(L.load().bx_adamw_step(_p(self.flat_p), _p(self.flat_g), _p(self.exp_avg), _p(self.exp_avg_sq), self.n, g["lr"],
                                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], float(self.grad_scale),
                                           _p(self.step_t), _stream()), "bx_adamw_step")
        else:  # host tensors: only reached by the gloo/CPU logic tests of the data-parallel wrapper
            self.step_t += 1
            t = float(self.step_t)
            b1, b2 = g["betas"]
            gr = self.flat_g * self.grad_scale
            self.flat_p.mul_(1 - g["lr"] * g["weight_decay"])
            self.exp_avg.mul_(b1).add_(gr, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(gr, gr, value=1 - b2)
            denom = (self.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(g["eps"])
            self.flat_p.addcdiv_(self.exp_avg, denom, value=-g["lr"] / (1 - b1 ** t))

    def state_dict(self):
        return {"flat_adamw": 1, "step": self.step_t.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        if "flat_adamw" not in sd:
            raise ValueError("FlatAdamW.load_state_dict: not a FlatAdamW state")
        self.step_t.copy_(sd["step"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]


# ------------------------------------------------------------------------------------------------
def setup(rank, world_size, backend=None):
    """dist.init_process_group (reference XAI_Multimodality.py:66-68); backend 'nccl' is RCCL on ROCm."""
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world_size)


def cleanup():
    if dist.is_initialized():
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


def create_ddp_model(model, device_ids=None, output_device=None):
    """reference XAI_Multimodality.py:77-80."""
    if device_ids:
        model = model.to(device_ids[0])
    return DataParallel(model, device_ids=device_ids, output_device=output_device)


# ------------------------------------------------------------------------------------------------
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

    A step is ~150 kernel launches; issued one by one from Python the host becomes the bottleneck (2.44 ms vs 2.0 ms per
    step on MI355X).  The first batch of a given shape runs eagerly, the second is captured, later ones are replayed:
    inputs are copied into the graph's static buffers, the dropout seeds and BatchNorm statistics live on the device and
    advance inside the graph.  Needs FlatAdamW (stable gradient arena) and CUDA inputs; anything else, or BX_GRAPH_LOOPS=0,
    falls back to the eager step.  Returned loss / output are the graph's static tensors (valid until the next call)."""

    def __init__(self, model, optimizer, criterion, ddp=None):
        self.model, self.opt, self.crit, self.ddp = model, optimizer, criterion, ddp
        self.enabled = isinstance(optimizer, FlatAdamW) and optimizer.is_cuda and os.environ.get("BX_GRAPH_LOOPS", "1") != "0"
        self._seen, self._graphs = set(), {}
        # branch-parallel replay (MultimodalModel only): see _capture_branches
        self.branches = os.environ.get("BX_BRANCH_GRAPHS", "0") == "1"     # off by default: see _capture_branches
        self._side = None

    def _capture_branches(self, static_in, static_lab):
        """The two-branch model as SIX graphs instead of one.  A replayed hipGraph runs its kernels strictly one after the
        other on this stack, even across forked branches (tools/graph_concurrency_probe.py), but two graph launches on two
        streams do overlap.  The EEG branch is ~25 small, latency-bound kernels forward and backward that are independent
        of the spectrogram branch until the fusion head, so the step is cut at the head:
            main stream:  [zero grads, spectrogram features] -> [head fwd, loss, head bwd] -> [spectrogram bwd] -> [AdamW]
            side stream:  [EEG features]                  ->            (join)           -> [EEG bwd]        -> (join)
        Autograd is cut at the two feature tensors (detached leaves feed the head; their .grad seeds the branch backward
        passes).  Every graph has its own memory pool: graphs that replay concurrently must not share one."""
        m = self.model
        em, sm = m.eeg_model, m.spectrogram_model
        eeg_in, spec_in = static_in
        cur = torch.cuda.current_stream()
        s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
        s_main.wait_stream(cur); s_side.wait_stream(cur)
        g = [torch.cuda.CUDAGraph() for _ in range(6)]
        with torch.cuda.graph(g[0], stream=s_main):
            self.opt.zero_grad()
            sf = sm.features(spec_in)
        with torch.cuda.graph(g[1], stream=s_side):
            ef = em.features(eeg_in)
        with torch.cuda.graph(g[2], stream=s_main):
            sf_leaf, ef_leaf = sf.detach().requires_grad_(True), ef.detach().requires_grad_(True)
            out = ops.MultimodalHeadFn.apply(sf_leaf.permute(0, 2, 3, 1), ef_leaf, sm.fc.weight, sm.fc.bias, em.dense.weight, em.dense.bias,
                                             m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
            loss = self.crit(out, static_lab)
            loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
            d_sf, d_ef = sf_leaf.grad, ef_leaf.grad
        with torch.cuda.graph(g[3], stream=s_main):
            torch.autograd.backward([sf], [d_sf])
        with torch.cuda.graph(g[4], stream=s_side):
            torch.autograd.backward([ef], [d_ef])
        with torch.cuda.graph(g[5], stream=s_main):
            if self.ddp is None:
                self.opt.step()
            else:
                self.opt.gather_grads()
        cur.wait_stream(s_main); cur.wait_stream(s_side)
        self._side = s_side
        ev = [torch.cuda.Event() for _ in range(4)]
        return ("branches", g, ev, static_in, static_lab, loss.detach(), out.detach(), (sf, ef, sf_leaf, ef_leaf, d_sf, d_ef))

    def _replay_branches(self, entry):
        _, g, ev, _, _, loss, out, _ = entry
        cur, side = torch.cuda.current_stream(), self._side
        if os.environ.get("BX_BRANCH_SYNC") == "1":         # debugging aid: the same six graphs with device-wide syncs between stages
            for k in (1, 0, 2, 4, 3, 5):
                with torch.cuda.stream(side if k in (1, 4) else cur):
                    g[k].replay()
                torch.cuda.synchronize()
            return loss, out
        ev[0].record(cur)                               # inputs copied, previous step's optimizer update done
        side.wait_event(ev[0])
        with torch.cuda.stream(side):
            g[1].replay()
            ev[1].record(side)
        g[0].replay()
        cur.wait_event(ev[1])
        g[2].replay()
        ev[2].record(cur)
        side.wait_event(ev[2])
        with torch.cuda.stream(side):
            g[4].replay()
            ev[3].record(side)
        g[3].replay()
        cur.wait_event(ev[3])
        g[5].replay()
        if self.ddp is not None:
            self.ddp.sync_gradients(self.opt)
            self.opt.step(gathered=True)
        return loss, out

    def _eager(self, inputs, labels):
        self.opt.zero_grad()
        out = self.model(*inputs)
        loss = self.crit(out, labels)
        loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
        if self.ddp is not None:
            self.ddp.sync_gradients(self.opt)
            self.opt.step(gathered=True)
        else:
            self.opt.step()
        return loss.detach(), out.detach()

    def __call__(self, inputs, labels):
        if not self.enabled or not self.model.training:
            return self._eager(inputs, labels)
        key = (tuple((tuple(t.shape), t.dtype) for t in inputs), tuple(labels.shape), labels.dtype)
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen:                 # first batch of this shape: plain step (also sizes every workspace)
                self._seen.add(key)
                return self._eager(inputs, labels)
            static_in = [t.detach().clone() for t in inputs]
            static_lab = labels.detach().clone()
            entry = None
            if self.branches and type(self.model).__name__ == "MultimodalModel" and len(inputs) == 2 and self.model._fusable():
                try:
                    torch.cuda.synchronize()
                    entry = self._capture_branches(static_in, static_lab)
                except Exception as exc:              # noqa: BLE001
                    print(f"[brainxai] branch-parallel capture failed ({type(exc).__name__}: {exc}); capturing one graph instead")
                    self.branches, entry = False, None
                    torch.cuda.synchronize()
                    self.opt.zero_grad()
            if entry is not None:
                self._graphs[key] = entry
                return self._replay_entry(entry, inputs, labels)
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.opt.zero_grad()
                    out = self.model(*static_in)
                    loss = self.crit(out, static_lab)
                    loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
                    if self.ddp is None:
                        self.opt.step()
                entry = (graph, static_in, static_lab, loss.detach(), out.detach())
            except Exception as exc:                  # noqa: BLE001  (capture is an optimisation, never a requirement)
                print(f"[brainxai] hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly")
                self.enabled = False
                torch.cuda.synchronize()
                return self._eager(inputs, labels)
            self._graphs[key] = entry
        return self._replay_entry(entry, inputs, labels)

    def _replay_entry(self, entry, inputs, labels):
        static_in, static_lab = (entry[3], entry[4]) if entry[0] == "branches" else (entry[1], entry[2])
        for dst, src in zip(static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if static_lab.data_ptr() != labels.data_ptr():
            static_lab.copy_(labels, non_blocking=True)
        if entry[0] == "branches":
            return self._replay_branches(entry)
        graph, _, _, loss, out = entry
        graph.replay()
        if self.ddp is not None:
            self.ddp.sync_gradients(self.opt)
            self.opt.step(gathered=True)
        return loss, out


_STEPPER = [None]


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

    def save(self, state):
        self.wait()                                        # one write in flight; also frees the pinned buffers for reuse
        if torch.cuda.is_available():
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                snap = self._snapshot(state, "")
                done = torch.cuda.Event()
                done.record()
        else:
            snap, done = self._snapshot(state, ""), None

        def work():
            try:
                if done is not None:
                    done.synchronize()
                cpu = _to_plain(snap)
                torch.save(cpu, self.path + ".tmp")
                os.replace(self.path + ".tmp", self.path)
            except Exception as exc:                       # noqa: BLE001  (surfaced by wait())
                self.error = exc
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


def _run_epoch(model, loader, criterion, device, optimizer=None, ddp=None, unpack=None):
    """One pass; loss*B and correct counts accumulate ON DEVICE (the reference syncs three times per step)."""
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    correct = torch.zeros((), dtype=torch.int64, device=device)
    total = 0
    stepper = None
    for batch in loader:
        inputs, labels = unpack(batch) if unpack else batch
        inputs = [t.to(device, non_blocking=True) for t in (inputs if isinstance(inputs, (tuple, list)) else (inputs,))]
        labels = labels.to(device, non_blocking=True)
        if optimizer is not None:
            if stepper is None:
                s = _STEPPER[0]                             # one cached stepper: captured graphs survive from epoch to epoch
                if s is not None and s.model is model and s.opt is optimizer and s.crit is criterion and s.ddp is ddp:
                    stepper = s
                else:
                    stepper = _STEPPER[0] = GraphedTrainStep(model, optimizer, criterion, ddp)
            loss, out = stepper(inputs, labels)
        else:
            with torch.no_grad():
                out = model(*inputs)
                loss = criterion(out, labels)
        n = labels.shape[0]
        loss_sum += loss.detach() * n
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
    for epoch in range(start, epochs):
        model.train()
        l, a = _run_epoch(model, train_loader, criterion, device, optimizer)
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


def train_and_validate_eeg_distributed(model, train_loader, valid_loader, epochs, optimizer, criterion, scheduler, device,
                                       checkpoint_dir, logger, rank, world_size):
    """Reference root/src/training/training_distributed.py:22-141 for a single-input model (EEGNet).
    The reference adds ``sum(p**2) * model.weight_decay`` to the loss (:52-53); here its value comes from one
    bx_sumsq launch and its gradient (2*wd*p) is added to the flat gradient arena before the all-reduce."""
    setup(rank, world_size)
    model = model.to(device)
    wd = float(getattr(model, "weight_decay", 0.0))
    ddp = DataParallel(model, device_ids=[rank] if torch.cuda.is_available() else None)
    name = "eeg_checkpoint.pth.tar"
    start, tr_l, va_l, tr_a, va_a = load_checkpoint(checkpoint_dir, name, ddp, optimizer)
    reg_losses = []
    flat = isinstance(optimizer, FlatAdamW)
    for epoch in range(start, epochs):
        if logger:
            logger.info(f"Starting Epoch {epoch + 1}/{epochs}")
        model.train()
        if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        loss_sum = torch.zeros((), device=device); reg_sum = torch.zeros((), device=device)
        correct = torch.zeros((), dtype=torch.int64, device=device); total = 0; nb = 0
        for data, labels in train_loader:
            data, labels = data.to(device), labels.to(device)
            optimizer.zero_grad()
            out = ddp(data)
            loss = criterion(out, labels)
            loss.backward(ops.unit_gradient(loss.device) if loss.is_cuda and loss.dim() == 0 else None)
            if wd > 0 and flat and optimizer.is_cuda:
                optimizer.gather_grads()
                reg = torch.empty((), dtype=torch.float32, device=device)
                L.check(L.load().bx_sumsq(_p(optimizer.flat_p), optimizer.n, _p(reg), _stream()), "bx_sumsq")
                L.check(L.load().bx_axpby(_p(optimizer.flat_p), _p(optimizer.flat_g), optimizer.n, 2.0 * wd, 1.0, _stream()), "bx_axpby")
                reg_sum += reg * wd
            elif wd > 0:
                with torch.no_grad():
                    for p in model.parameters():
                        if p.grad is not None:
                            p.grad.add_(p, alpha=2.0 * wd)
                            reg_sum += (p.detach() ** 2).sum() * wd
            ddp.sync_gradients(optimizer)
            optimizer.step(gathered=True) if flat else optimizer.step()
            loss_sum += loss.detach(); nb += 1
            correct += (out.detach().argmax(1) == labels.argmax(1)).sum(); total += labels.shape[0]
        tr_l.append(float(loss_sum) / max(nb, 1)); tr_a.append(float(correct) / max(total, 1) * 100.0)
        reg_losses.append(float(reg_sum) / max(nb, 1))
        model.eval()
        l, a = _run_epoch(ddp, valid_loader, criterion, device)
        va_l.append(l); va_a.append(a)
        if scheduler is not None:
            scheduler.step()
        if logger:
            logger.info(f"Epoch {epoch + 1}/{epochs} train {tr_l[-1]:.4f}/{tr_a[-1]:.2f}% valid {va_l[-1]:.4f}/{va_a[-1]:.2f}%")
        if rank == 0:
            save_checkpoint({"epoch": epoch + 1, "state_dict": ddp.state_dict(), "optimizer": optimizer.state_dict(),
                             "train_losses": tr_l, "valid_losses": va_l, "train_accuracies": tr_a, "valid_accuracies": va_a,
                             "lr_scheduler": scheduler.state_dict() if scheduler is not None else [],
                             "regularization_losses": reg_losses}, checkpoint_dir, name)
    return tr_l, va_l, tr_a, va_a
