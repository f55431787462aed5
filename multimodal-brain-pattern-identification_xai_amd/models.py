"""Drop-in model classes: same names, constructor arguments, sub-module names and state_dict keys as
the reference (root/src/models/models.py:42-107,239-289; MultimodalModel = XAI_Multimodality.py:1082-1108),
with forward/backward running on the HIP kernels of libbrainxai.so.

Sub-modules such as ``conv1`` or ``bn`` are ordinary torch modules used as PARAMETER CONTAINERS (so
checkpoints interchange with the reference and ``model.block3.conv2.weight`` means what it says);
their own ``forward`` is never called -- each stage runs as one fused autograd Function.

Inputs must be CUDA tensors; there is no CPU path.  ``compute_dtype`` (float32 | bfloat16) selects the
activation storage type; parameters, statistics and gradients stay float32.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops

_STAGES = ((16, "max"), (32, "avg"), (64, "max"), (128, "avg"), (256, "max"))


def _is_internal(x: torch.Tensor, dtype: torch.dtype) -> bool:
    """True for a logical-NCHW view of one of our channels-last activations."""
    return (x.is_cuda and x.dim() == 4 and x.dtype == dtype and x.shape[1] % 8 == 0
            and x.permute(0, 2, 3, 1).is_contiguous())


class Block(nn.Module):
    """relu(conv3x3) x3 -> 2x2 pool -> BatchNorm -> Dropout -> + conv1x1(bilinear(x))  (reference models.py:42-77)."""

    def __init__(self, in_channels, out_channels, pool_type="max", pool_size=(2, 2), dropout_p=0.5):
        super().__init__()
        if tuple(pool_size) != (2, 2) if not isinstance(pool_size, int) else pool_size != 2:
            raise ValueError("brainxai Block: only the reference's 2x2 pooling is implemented")
        if pool_type not in ("max", "avg"):
            raise ValueError(f"pool_type must be 'max' or 'avg', got {pool_type!r}")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv3 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.pool = nn.MaxPool2d(kernel_size=pool_size) if pool_type == "max" else nn.AvgPool2d(kernel_size=pool_size)
        self.bn = nn.BatchNorm2d(out_channels)
        self.dropout = nn.Dropout(p=dropout_p)
        self.conv1x1 = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.in_channels, self.out_channels, self.pool_type = in_channels, out_channels, pool_type
        self.compute_dtype = torch.float32
        self.salt = 0
        self._preact = 0          # set by explain.grad_cam for targets "blockN.convK"
        self._capture = None
        self._keep = None         # set by ops.keep_block_activations (parity tooling)
        self._prepacked, self._pack_base = None, 0      # set per forward by Spectrogram_Model (one pack launch for all stages)
        self._seed = None                               # idem: one dropout seed launch for all stages
        self._sync = None                               # device words of the in-launch statistics finalizes (bxTailDesc.sync)

    def _sync_words(self, device):
        """zeroed counter words owned by this block: the library's kernels count arrivals in them and leave them zero"""
        if not ops.TAIL_IN_LAUNCH:
            return None
        if self._sync is None or self._sync.device != device:
            self._sync = torch.zeros(ops.L.BX_TAIL_SYNC_WORDS, dtype=torch.int32, device=device)
        return self._sync

    def forward(self, x):
        dt = self.compute_dtype
        if _is_internal(x, dt) and x.shape[1] == ops.pad8(self.in_channels):
            xi = x.permute(0, 2, 3, 1)
        else:
            if x.dim() != 4 or x.shape[1] != self.in_channels:
                raise RuntimeError(f"Block expected [B,{self.in_channels},H,W], got {tuple(x.shape)}")
            xi = ops.InputLayout.apply(x, dt)
        if xi.shape[1] < 2 or xi.shape[2] < 2:
            raise RuntimeError("Block needs H, W >= 2")
        bn = self.bn
        cfg = ops.block_cfg(pool=self.pool_type, training=self.training, dropout_p=self.dropout.p if self.training else 0.0,
                            eps=bn.eps, momentum=0.1 if bn.momentum is None else bn.momentum, salt=self.salt,
                            preact=self._preact, capture=self._capture, prepacked=self._prepacked, pack_base=self._pack_base, seed=self._seed,
                            keep=self._keep, sync=self._sync_words(xi.device), grad_mode=torch.is_grad_enabled())
        out = ops.BlockFn.apply(xi, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, self.conv3.weight,
                                self.conv3.bias, bn.weight, bn.bias, self.conv1x1.weight, self.conv1x1.bias,
                                bn.running_mean, bn.running_var, bn.num_batches_tracked, cfg)
        return out.permute(0, 3, 1, 2)     # logical NCHW view (channels-last strides), as hooks expect


class Spectrogram_Model(nn.Module):
    """Five stages -> GAP -> Linear(256, classes) -> LogSoftmax (reference models.py:79-107).
    ``in_channels`` (default 3, the reference's value) is the build's extension for the 4-plane benchmark input."""

    def __init__(self, num_classes=6, in_channels=3):
        super().__init__()
        prev = in_channels
        for i, (c, kind) in enumerate(_STAGES, start=1):
            blk = Block(in_channels=prev, out_channels=c, pool_type=kind, pool_size=(2, 2))
            blk.salt = i
            setattr(self, f"block{i}", blk)
            prev = c
        self.gap = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(256, num_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)
        self.compute_dtype = torch.float32

    def _pack_all(self, x=None, seed_pair=False):
        """bf16 MFMA path: ONE launch packs the forward and data-gradient operands of all 15 convolutions -- and, when the raw
        batch ``x`` (fp32 NCHW, no gradient wanted) is given, converts it to the internal layout as well (returned, else None).
        ``seed_pair=True`` (MultimodalModel's training forward): the same launch advances both branches' dropout counters and the
        return value is (xi, spectrogram seed, EEG seed); without the MFMA path the seeds come from their own launch."""
        blocks = [getattr(self, f"block{i}") for i in range(1, 6)]
        for blk in blocks:
            blk._prepacked = None
        if ops.CONV_ALGO == ops.L.BX_ALGO_DIRECT or not self.fc.weight.is_cuda:
            return (None,) + tuple(ops.next_seed_pair(x.device)) if seed_pair else None
        weights = [getattr(b, f"conv{k}").weight for b in blocks for k in (1, 2, 3)]
        plan = getattr(self, "_pack_plan", None)
        if plan is None or plan.key != tuple(w.data_ptr() for w in weights) + (self.compute_dtype,):
            plan = ops.PackPlan(weights, self.compute_dtype)     # fp32 storage: split (hi + lo) operands, same single launch
            self._pack_plan = plan
        raw = (x is not None and x.is_cuda and x.dim() == 4 and x.shape[1] == blocks[0].in_channels and not x.requires_grad
               and not _is_internal(x, torch.bfloat16) and blocks[0].compute_dtype == torch.bfloat16)
        if seed_pair and plan.njobs:
            res = plan.run_step(x if raw else None)
        elif seed_pair:
            res = (plan.run(x if raw else None),) + tuple(ops.next_seed_pair(x.device))
        else:
            res = plan.run(x if raw else None, reuse=not self.training and ops.PACK_REUSE[0])     # frozen-weight scopes: no repack while the weights stand
        for bi, blk in enumerate(blocks):
            blk._prepacked, blk._pack_base = plan, 3 * bi
        return res

    def features(self, x, seed=None, cut=None, packed=None):
        """``cut = (k, fn)``: the activation leaving stage k is handed to ``fn`` and its return value feeds stage k+1 -- the
        data-parallel step uses it to cut autograd there (fn detaches), so that the gradients of the late stages (88 % of the
        parameter bytes, finished first) can be all-reduced while the early stages' backward still runs.
        ``packed``: the caller already ran ``_pack_all`` for this forward pass and hands over its result (a 1-tuple)."""
        xi = packed[0] if packed is not None else self._pack_all(x)
        if xi is not None:
            x = xi                                          # already in the internal layout (same launch as the weight packing)
        blocks = [getattr(self, f"block{i}") for i in range(1, 6)]
        if self.training and x.is_cuda and any(b.dropout.p > 0 for b in blocks):
            if seed is None:
                seed = ops.next_seed(x.device)
            for b in blocks:
                b._seed = seed
        try:
            for i, b in enumerate(blocks, start=1):
                x = b(x)
                if cut is not None and cut[0] == i:
                    x = cut[1](x)
        finally:
            for b in blocks:
                b._prepacked, b._seed = None, None
        return x

    def forward(self, x):
        f = self.features(x)
        return ops.GapFcLsmFn.apply(f.permute(0, 2, 3, 1), self.fc.weight, self.fc.bias)


class EEGNet(nn.Module):
    """EEGNet over [B,1,Chans,Samples] (reference models.py:239-289)."""

    def __init__(self, nb_classes, Chans=37, Samples=3000, dropoutRate=0.5, kernLength=64, F1=8, D=2, F2=16,
                 norm_rate=0.25, dropoutType="Dropout"):
        super().__init__()
        if dropoutType not in ("Dropout", "Dropout2d"):
            raise ValueError("dropoutType must be 'Dropout' or 'Dropout2d' (reference models.py:255)")
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
        self._geom = SimpleNamespace(F1=F1, D=D, F2=F2, K1=kernLength, K2=16, P1=4, P2=8)
        self.compute_dtype = torch.float32
        self.salt = 100 | (0x80000000 if dropoutType == "Dropout2d" else 0)     # bit 31: channel-wise mask (nn.Dropout2d)

    def features(self, x, seed=None):
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.Chans:
            raise RuntimeError(f"EEGNet expected [B,1,{self.Chans},T], got {tuple(x.shape)}")
        g = self._geom
        bn1, bn2, bn3 = self.batchnorm1, self.batchnorm2, self.batchnorm3
        cfg = SimpleNamespace(F1=g.F1, D=g.D, F2=g.F2, K1=g.K1, K2=g.K2, P1=g.P1, P2=g.P2, training=self.training,
                              eps=bn1.eps, momentum=0.1 if bn1.momentum is None else bn1.momentum,
                              dropout_p=self.dropout.p if self.training else 0.0, salt=self.salt, dtype=self.compute_dtype, seed=seed,
                              grad_mode=torch.is_grad_enabled())      # (inside an autograd Function's forward grad mode always reads False)
        bufs = (bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, bn2.running_mean, bn2.running_var,
                bn2.num_batches_tracked, bn3.running_mean, bn3.running_var, bn3.num_batches_tracked)
        return ops.EegFeaturesFn.apply(x, self.conv1.weight, bn1.weight, bn1.bias, self.depthwiseConv.weight, bn2.weight, bn2.bias,
                                       self.separableConv.weight, bn3.weight, bn3.bias, bufs, cfg)

    def forward(self, x):
        feat = self.features(x)
        if feat.shape[1] != self.dense.in_features:
            raise RuntimeError(f"EEGNet: {feat.shape[1]} features but dense expects {self.dense.in_features} (Samples mismatch)")
        return ops.LinearLsmFn.apply(feat, self.dense.weight, self.dense.bias)


class Attention(nn.Module):
    """Single-head scaled dot-product self-attention (reference models.py:109-134).  Inside ``EEGNetAttentionDeep`` it is a
    parameter container (the attention runs in the fused head kernels); called on its own it takes tokens [B, L, input_dim]
    and returns (output, attention_weights) like the reference (``bx_attention_fwd/_bwd``; input_dim = attention_dim = 32,
    L <= 32)."""

    def __init__(self, input_dim, attention_dim):
        super().__init__()
        self.query = nn.Linear(input_dim, attention_dim)
        self.key = nn.Linear(input_dim, attention_dim)
        self.value = nn.Linear(input_dim, attention_dim)
        self.scale = attention_dim ** -0.5

    def forward(self, x):
        return ops.AttentionFn.apply(x, self.query.weight, self.query.bias, self.key.weight, self.key.bias, self.value.weight, self.value.bias)


class EEGNetAttentionDeep(nn.Module):
    """EEGNet blocks 1-2, a third temporal block and self-attention over time (reference models.py:136-235).
    Blocks 1-2 run through the same kernels as ``EEGNet``; everything from ``conv2`` on is ``ops.EegDeepHeadFn``.
    ``last_attention`` holds the [B, L, L] softmax weights of the latest forward (the reference discards them)."""

    def __init__(self, nb_classes, Chans=37, Samples=3000, dropoutRate=0.5, kernLength=64, F1=8, D=2, F2=16, F3=32,
                 norm_rate=0.25, dropoutType="Dropout"):
        super().__init__()
        if dropoutType not in ("Dropout", "Dropout2d"):
            raise ValueError("dropoutType must be 'Dropout' or 'Dropout2d' (reference models.py:152-164)")
        _Drop = nn.Dropout if dropoutType == "Dropout" else nn.Dropout2d
        self.nb_classes, self.Chans, self.Samples = nb_classes, Chans, Samples
        self.conv1 = nn.Conv2d(1, F1, (1, kernLength), padding="same", bias=False)
        self.batchnorm1 = nn.BatchNorm2d(F1)
        self.depthwiseConv = nn.Conv2d(F1, F1 * D, (Chans, 1), groups=F1, bias=False)
        self.batchnorm2 = nn.BatchNorm2d(F1 * D)
        self.activation = nn.ELU()
        self.avg_pool1 = nn.AvgPool2d((1, 4))
        self.dropout1 = _Drop(dropoutRate)
        self.separableConv = nn.Conv2d(F1 * D, F2, (1, 16), padding="same", bias=False)
        self.batchnorm3 = nn.BatchNorm2d(F2)
        self.avg_pool2 = nn.AvgPool2d((1, 8))
        self.dropout2 = _Drop(dropoutRate)
        self.conv2 = nn.Conv2d(F2, F3, (1, 16), padding="same", bias=False)
        self.batchnorm4 = nn.BatchNorm2d(F3)
        self.avg_pool3 = nn.AvgPool2d((1, 8))
        self.dropout3 = _Drop(dropoutRate)
        self.attention_layer = Attention(F3, F3)
        self.output_samples = ((Samples // 4) // 8) // 8
        self.flattened_size = F3 * self.output_samples
        self.flatten = nn.Flatten()
        self.dense1 = nn.Linear(self.flattened_size, 128)
        self.dense2 = nn.Linear(128, nb_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)
        self._geom = SimpleNamespace(F1=F1, D=D, F2=F2, F3=F3, K1=kernLength, K2=16, K3=16, P1=4, P2=8, P3=8)
        self.compute_dtype = torch.float32
        self.salt = 200 | (0x80000000 if dropoutType == "Dropout2d" else 0)     # bit 31: channel-wise masks (nn.Dropout2d)
        self.last_attention = None

    def features(self, x):
        """Block 1-2 output after dropout2, flattened [B, F2 * (T//32)] (same kernels as EEGNet.features)."""
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.Chans:
            raise RuntimeError(f"EEGNetAttentionDeep expected [B,1,{self.Chans},T], got {tuple(x.shape)}")
        g = self._geom
        bn1, bn2, bn3 = self.batchnorm1, self.batchnorm2, self.batchnorm3
        cfg = SimpleNamespace(F1=g.F1, D=g.D, F2=g.F2, K1=g.K1, K2=g.K2, P1=g.P1, P2=g.P2, training=self.training,
                              eps=bn1.eps, momentum=0.1 if bn1.momentum is None else bn1.momentum,
                              dropout_p=self.dropout1.p if self.training else 0.0, salt=self.salt, dtype=self.compute_dtype,
                              dropout_p2=self.dropout2.p if self.training else 0.0,      # the second dropout's own rate (models.py:158,163)
                              grad_mode=torch.is_grad_enabled())
        bufs = (bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, bn2.running_mean, bn2.running_var,
                bn2.num_batches_tracked, bn3.running_mean, bn3.running_var, bn3.num_batches_tracked)
        return ops.EegFeaturesFn.apply(x, self.conv1.weight, bn1.weight, bn1.bias, self.depthwiseConv.weight, bn2.weight, bn2.bias,
                                       self.separableConv.weight, bn3.weight, bn3.bias, bufs, cfg)

    def forward(self, x):
        feat = self.features(x)
        g, bn4, att = self._geom, self.batchnorm4, self.attention_layer
        T2 = feat.shape[1] // g.F2
        if g.F3 * (T2 // g.P3) != self.dense1.in_features:
            raise RuntimeError(f"EEGNetAttentionDeep: {g.F3 * (T2 // g.P3)} features but dense1 expects {self.dense1.in_features} (Samples mismatch)")
        cfg = SimpleNamespace(T2=T2, F2=g.F2, F3=g.F3, K3=g.K3, P3=g.P3, training=self.training, eps=bn4.eps,
                              momentum=0.1 if bn4.momentum is None else bn4.momentum,
                              dropout_p=self.dropout3.p if self.training else 0.0, salt=self.salt + 1)
        logp, attn = ops.EegDeepHeadFn.apply(feat, self.conv2.weight, bn4.weight, bn4.bias, att.query.weight, att.query.bias,
                                             att.key.weight, att.key.bias, att.value.weight, att.value.bias, self.dense1.weight,
                                             self.dense1.bias, self.dense2.weight, self.dense2.bias,
                                             (bn4.running_mean, bn4.running_var, bn4.num_batches_tracked), cfg)
        self.last_attention = attn
        return logp


class MultimodalModel(nn.Module):
    """Late fusion over the two branches' log-probabilities (reference XAI_Multimodality.py:1082-1108)."""

    def __init__(self, eeg_model, spectrogram_model, num_classes=6):
        super().__init__()
        self.eeg_model = eeg_model
        self.spectrogram_model = spectrogram_model
        combined_output_size = eeg_model.dense.out_features + spectrogram_model.fc.out_features
        self.fc1 = nn.Linear(combined_output_size, 128)
        self.fc2 = nn.Linear(128, num_classes)
        self.log_softmax = nn.LogSoftmax(dim=1)

    def forward(self, eeg_data, spectrogram_data, cut=None):
        """``cut``: see Spectrogram_Model.features (build extension used by the overlapped data-parallel step; the reference's
        signature is the two positional inputs, XAI_Multimodality.py:1095)."""
        if cut is not None:
            if not self._fusable():
                raise RuntimeError("brainxai MultimodalModel: cut= needs the un-hooked reference architecture (fused head path)")
            em, sm = self.eeg_model, self.spectrogram_model
            ss = se = packed = None
            if self.training and em.dropout.p > 0 and any(getattr(sm, f"block{i}").dropout.p > 0 for i in range(1, 6)):
                xi, ss, se = sm._pack_all(spectrogram_data, seed_pair=True)
                packed = (xi,)
            if eeg_data.is_cuda and ops.OVERLAP_EEG_DDP and ops.overlap_eeg_now():      # (opt-in; see the fused path below and ops.overlap_eeg_now)
                cur, side = ops.fork_eeg(eeg_data.device, se, eeg_data)
                with torch.cuda.stream(side):
                    ef = em.features(eeg_data, seed=se)
                sf = sm.features(spectrogram_data, seed=ss, cut=cut, packed=packed)
                cur.wait_stream(side)
                ef.record_stream(cur)
            else:
                ef = em.features(eeg_data, seed=se)
                sf = sm.features(spectrogram_data, seed=ss, cut=cut, packed=packed)
            return ops.MultimodalHeadFn.apply(sf.permute(0, 2, 3, 1), ef, sm.fc.weight, sm.fc.bias, em.dense.weight, em.dense.bias,
                                              self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        if ops.OVERLAP and eeg_data.is_cuda and ops.CONV_PROFILE is None:
            # the EEG branch (small, latency-bound kernels) runs on a side stream beside the spectrogram branch; autograd
            # replays each branch's backward on the stream its forward ran on, so the backward overlaps too
            cur = torch.cuda.current_stream()
            side = ops.side_stream("eeg", eeg_data.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                e = self.eeg_model(eeg_data)
            s = self.spectrogram_model(spectrogram_data)
            cur.wait_stream(side)
            e.record_stream(cur)
        elif self._fusable():
            # one launch for GAP+fc, dense and the fusion head (same arithmetic as the three separate ops below)
            em, sm = self.eeg_model, self.spectrogram_model
            ss = se = packed = None
            if self.training and em.dropout.p > 0 and any(getattr(sm, f"block{i}").dropout.p > 0 for i in range(1, 6)):
                # both branches' dropout seeds ride in the weight-packing launch, which therefore opens the step
                xi, ss, se = sm._pack_all(spectrogram_data, seed_pair=True)
                packed = (xi,)
            if eeg_data.is_cuda and ops.overlap_eeg_now():
                # Round 3: the EEG branch (twenty small-grid, latency-bound launches, ~180 us of the step) runs on a side stream beside
                # the spectrogram branch -- ONE fork after the packing launch (its seeds feed both branches) and one join in front of
                # the head; autograd replays the branch's backward on the stream its forward ran on, so the backward overlaps the same
                # way.  Inside a captured step these are two parallel chains of the graph.
                cur, side = ops.fork_eeg(eeg_data.device, se, eeg_data)
                with torch.cuda.stream(side):
                    ef = em.features(eeg_data, seed=se)
                sf = sm.features(spectrogram_data, seed=ss, packed=packed)
                cur.wait_stream(side)
                ef.record_stream(cur)
            else:
                ef = em.features(eeg_data, seed=se)
                sf = sm.features(spectrogram_data, seed=ss, packed=packed)
            if ef.shape[1] != em.dense.in_features:
                raise RuntimeError(f"EEGNet: {ef.shape[1]} features but dense expects {em.dense.in_features} (Samples mismatch)")
            return ops.MultimodalHeadFn.apply(sf.permute(0, 2, 3, 1), ef, sm.fc.weight, sm.fc.bias, em.dense.weight, em.dense.bias,
                                              self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        else:
            e = self.eeg_model(eeg_data)
            s = self.spectrogram_model(spectrogram_data)
        if e.shape[1] != s.shape[1]:
            raise RuntimeError("MultimodalModel: both branches must emit the same number of classes")
        return ops.FusionHeadFn.apply(e, s, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)

    def _fusable(self) -> bool:
        """The fused head skips the branch modules' own forward(): only when nobody hooked them and the sizes fit its kernels."""
        em, sm = self.eeg_model, self.spectrogram_model
        if not ops.FUSED_HEAD or type(em) is not EEGNet or type(sm) is not Spectrogram_Model:
            return False
        if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in (em, sm, em.dense, sm.fc)):
            return False
        n = self.fc2.out_features
        return (em.dense.out_features == sm.fc.out_features == n and self.fc1.in_features == 2 * n and n <= 32
                and self.fc1.out_features <= 256 and self.fc1.out_features * 2 * n <= 4096 and em.dense.in_features <= 4096
                and sm.fc.in_features <= 1024)

    def forward_spectrogram(self, spectrogram_data):
        return self.spectrogram_model(spectrogram_data)


class KLDivLoss(nn.Module):
    """nn.KLDivLoss(reduction) on log-prob input / prob target, fused with its gradient seed."""

    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction not in ("mean", "batchmean", "sum"):
            raise ValueError(reduction)
        self.reduction = reduction

    def forward(self, log_probs, target):
        return ops.KLDivFn.apply(log_probs, target, self.reduction, 1.0)


def set_compute_dtype(model: nn.Module, dtype: torch.dtype) -> nn.Module:
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    for m in model.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    return model


def build_multimodal(chans=19, samples=2000, in_channels=4, num_classes=6, dropout=0.5, compute_dtype=torch.float32):
    """The benchmark model of BASELINE.md section 2 (2 025 074 parameters at the defaults)."""
    eeg = EEGNet(num_classes, Chans=chans, Samples=samples, dropoutRate=dropout)
    spec = Spectrogram_Model(num_classes, in_channels=in_channels)
    for i in range(1, 6):
        getattr(spec, f"block{i}").dropout.p = dropout
    return set_compute_dtype(MultimodalModel(eeg, spec, num_classes), compute_dtype)
