"""GPU parity AT THE BENCHMARKED CONFIGURATION (BASELINE.json configs[0] and configs[1]): exactly what bench.py times --
B=64, spectrograms [64,4,128,256] + raw EEG [64,10000,19] through the stacker -> [64,1,19,2000], train mode -- is held to the
oracle here (dropout 0: the two sides cannot share a Bernoulli stream).

* fp32 storage: logits, loss, every parameter gradient and the block5 Grad-CAM maps of all 6 classes within 1e-3 (north_star).
* bf16 storage (the benchmark dtype): the SAME numbers on both sides -- inputs and every conv weight rounded to bf16-exact
  values first -- so that the only difference is bf16 rounding of stored activations / activation gradients, with bounds
  derived below instead of guessed.
* configs[0]: the spectrogram CNN alone on 32 x [4,128,256], forward + backward + the debug epoch (2 AdamW steps of B=16).

The oracle runs in fp32 on the host cores of the GPU box (one forward+backward at B=64 takes a few seconds).
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import brainxai
from brainxai import ops
from oracle import ref_torch as O
from tests.golden_util import check_decisions, grad_close, matched_oracle, rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-3
B, CIN, H, W, CHANS, T = 64, 4, 128, 256, 19, 2000


def _bench_inputs(bf16_exact=False):
    """bench.py's synthetic batch: spec U[0,1), raw EEG N(0,100^2) with NaNs and outliers -> GPU stacker."""
    syn = O.synthetic_batch(batch=B, in_channels=CIN, height=H, width=W, chans=CHANS, seed=42, stacked=False)
    eeg = brainxai.stack_eeg(syn["raw_eeg"].to(DEV)).cpu()          # the product's stacker feeds both sides (its own parity: test_stacker)
    spec, labels = syn["spec"], syn["labels"]
    if bf16_exact:
        eeg, spec = eeg.bfloat16().float(), spec.bfloat16().float()
    return eeg, spec, labels


def _models(seed, dtype, bf16_exact=False):
    ref = O.fill_params(O.build_multimodal(CHANS, T, CIN, dropout=0.0), seed=seed)
    if bf16_exact:
        with torch.no_grad():
            for n, p in ref.named_parameters():
                if p.dim() == 4:                                        # every convolution weight (both branches)
                    p.copy_(p.bfloat16().float())
    mine = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.0, compute_dtype=dtype)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.to(DEV)


def _gscale(model):
    return 1e-2 * max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)


def test_bench_config_fp32_train_step_and_gradcam():
    """configs[1] shapes, fp32 storage, train mode.  Logits and loss within 1e-3 of the fp32 oracle (north_star).  Parameter
    gradients: at this size every forward contains a few dozen ReLU pre-activations within 1e-8 of zero, and the reference's OWN
    fp32 gradients sit 1.4e-3 from their exact (fp64) values (printed below).  The gradient target is therefore the fp64 oracle
    with the HIP forward's decisions pinned (tests/golden_util.matched_oracle: every decision that differs from the exact forward
    must be a demonstrated tie) -- and against that target the comparison is strict, every tensor within 1e-3.  Then eval mode:
    block5 Grad-CAM maps of all 6 classes (configs[3]'s map) within 1e-3 of the raw-map scale."""
    ref, mine = _models(5, torch.float32)
    eeg, spec, labels = _bench_inputs()
    ref.train(); mine.train()
    ref64 = copy.deepcopy(ref).double()
    out_r = ref(eeg, spec)
    loss_r = O.kl_div(out_r, labels)
    loss_r.backward()
    O.kl_div(ref64(eeg.double(), spec.double()), labels.double()).backward()
    try:
        keep = ops.keep_block_activations(mine)
        out = mine(eeg.to(DEV), spec.to(DEV))
        loss = brainxai.KLDivLoss()(out, labels.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        e_out = rel_err(out.detach().cpu(), out_r.detach())
        assert e_out < TOL
        assert abs(float(loss.detach()) - float(loss_r.detach())) <= TOL * abs(float(loss_r.detach()))
        twin, flips = matched_oracle(O, ref, (eeg, spec), keep, "bench config fp32")
        ops.keep_block_activations(mine, on=False)
        del keep
        O.kl_div(twin(eeg.double(), spec.double()), labels.double()).backward()
        fl = 1e-2 * max(float(p.grad.abs().max()) for p in twin.parameters())
        worst, worst_ref = 0.0, 0.0
        for (n, p), (_, q32), (_, q64), (_, q) in zip(mine.named_parameters(), ref.named_parameters(), ref64.named_parameters(), twin.named_parameters()):
            worst = max(worst, grad_close(p.grad.cpu(), q.grad, TOL, label=f"bench fp32 d{n} (vs decision-matched fp64 oracle)", floor=fl))
            worst_ref = max(worst_ref, rel_err(q32.grad, q64.grad, floor=fl))
        print(f"[parity] bench config fp32: logits {e_out:.2e}; parameter gradients: HIP path vs the decision-matched fp64 oracle {worst:.2e} "
              f"({len(flips)} demonstrated tie flips pinned); for scale, the fp32 oracle vs its own fp64 run: {worst_ref:.2e}")
        for (n, t), (_, t2) in zip(mine.named_buffers(), ref.named_buffers()):
            assert rel_err(t.float().cpu(), t2.float()) < TOL, n         # BatchNorm running statistics after the step
    finally:
        ops.keep_block_activations(mine, on=False)
        ops.clear_grad_views()
    # Grad-CAM at the benchmark's target on a quarter of the batch (the CPU hook version costs ~1 s per 8 samples per class)
    cam = brainxai.grad_cam(mine, eeg[:16].to(DEV), spec[:16].to(DEV), class_idx="all")
    want = O.grad_cam(ref, eeg[:16], spec[:16], class_idx="all")
    raw = O.grad_cam(ref, eeg[:16], spec[:16], class_idx="all", relu=False)
    torch.cuda.synchronize()
    assert cam.shape == want.shape == (16, 6, H, W)
    assert rel_err(cam.cpu(), want, floor=float(raw.abs().max())) < TOL


def test_bench_config_bf16_train_step():
    """configs[1] exactly as bench.py runs it (bf16 storage, MFMA kernels, B=64, train mode; inputs and convolution weights
    bf16-exact on both sides so that operand conversion is not part of the comparison).

    What bf16 storage does to THIS workload is not a kernel property, and it is measured here instead of being waved at.  The
    3x3-convolution gradients of a randomly initialised network on noise inputs are small residuals: each stage ends in a
    train-mode BatchNorm, which projects the per-channel mean and the component along x-hat out of the gradient that flows back
    into the convolutions, and what is left is a sum of 10^4..10^6 incoherent terms.  Rounding every stored activation to 8
    significant bits moves ~1 % of the ReLU decisions of the next layer, and through 15 layers that changes those residual sums
    by 20-50 % of their norm (direction cosine 0.88-0.98) -- on ANY implementation: the oracle's own algorithm with its stored
    tensors rounded to bf16 (oracle.ref_torch.decision_matched_twin(storage=bfloat16), CPU, fp64 arithmetic) sits exactly there
    against the fp32 oracle, while the 1x1 skip convolutions, BatchNorm affines and heads, whose gradients are coherent sums,
    agree to 2e-3..6e-2.  Both numbers are printed below.  A cosine >= 0.99 against the fp32 oracle is therefore not a property
    any bf16-storage implementation can have here; the strict statement that CAN be made, and is asserted, is:

      * against the bf16-STORAGE restatement of the oracle with the HIP forward's decisions pinned, every gradient tensor of
        >= 1024 entries has cosine >= 0.998 and a relative L2 error below ONE TENTH of that restatement's own distance from the
        fp32 oracle, and every pinned decision the restatement would have taken differently is a tie at bf16 resolution
        (margin < 2^-8).  Where the tenth comes from: the two sides accumulate the same bf16 operands in different orders (fp32
        MFMA tiles vs fp64), so a sum lands on the other side of a bf16 rounding boundary with probability
        p ~ (3e-7 S/|y|) / 2^-8 ~ 1.5e-4 per stored element, a full ulp (3.5x the rms rounding error) each: relative noise power
        p * 3.5^2 = 1.8e-3 of what rounding EVERY element injects, i.e. 4 % of its amplitude -- measured 3.7e-2 / 0.51 = 7 %;
      * against the fp32 oracle: logits within 2e-2 of their scale and the loss within 1e-2 (both are coherent quantities: 15
        roundings of rms 2^-9/sqrt(3) in sequence give 4.4e-3)."""
    ref, mine = _models(5, torch.bfloat16, bf16_exact=True)
    eeg, spec, labels = _bench_inputs(bf16_exact=True)
    ref.train(); mine.train()
    out_r = ref(eeg, spec)
    loss_r = O.kl_div(out_r, labels)
    loss_r.backward()
    try:
        keep = ops.keep_block_activations(mine)
        out = mine(eeg.to(DEV), spec.to(DEV))
        loss = brainxai.KLDivLoss()(out, labels.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        e_out = rel_err(out.detach().cpu(), out_r.detach())
        e_loss = abs(float(loss.detach()) - float(loss_r.detach())) / abs(float(loss_r.detach()))
        assert e_out < 2e-2, e_out
        assert e_loss < 1e-2, e_loss
        twin, _ = matched_oracle(O, ref, (eeg, spec), keep, "bench config bf16", storage=torch.bfloat16)
        ops.keep_block_activations(mine, on=False)
        del keep
        out_t = twin(eeg.double(), spec.double())
        n_dec, margin = check_decisions(twin, 2.0 ** -8, "bench config bf16")
        O.kl_div(out_t, labels.double()).backward()
        e_out_t = rel_err(out.detach().cpu(), out_t.detach())
        worst_cos, worst_l2, base_cos, base_l2, worst_n = 1.0, 0.0, 1.0, 0.0, ""
        table = []
        for (n, p), (_, q32), (_, q) in zip(mine.named_parameters(), ref.named_parameters(), twin.named_parameters()):
            if q.numel() < 1024:
                continue
            a, b, c = p.grad.flatten().cpu().double(), q.grad.flatten().double(), q32.grad.flatten().double()
            cos, l2 = float(F.cosine_similarity(a, b, dim=0)), float((a - b).norm() / b.norm())
            table.append((l2, cos, n))
            if l2 > worst_l2:
                worst_n = n
            worst_cos, worst_l2 = min(worst_cos, cos), max(worst_l2, l2)
            base_cos, base_l2 = min(base_cos, float(F.cosine_similarity(b, c, dim=0))), max(base_l2, float((b - c).norm() / c.norm()))
        print(f"[parity] bench config bf16: vs fp32 oracle: logits {e_out:.2e}, loss {e_loss:.2e}.  vs the bf16-storage restatement of the oracle "
              f"(decisions pinned; {n_dec} of them ties, largest margin {margin:.1e}): logits {e_out_t:.2e}, worst gradient cosine {worst_cos:.5f}, "
              f"worst rel L2 {worst_l2:.2e} ({worst_n}).  For scale, that restatement vs the fp32 oracle: worst cosine {base_cos:.3f}, worst rel L2 {base_l2:.2f}")
        for l2, cos, n in sorted(table, reverse=True)[:12]:
            print(f"[parity]    {n:48s} rel L2 {l2:.2e}  cosine {cos:.5f}")
        assert e_out_t < 5e-3, e_out_t
        assert worst_cos >= 0.998 and worst_l2 <= 0.1 * base_l2, (worst_n, worst_cos, worst_l2, base_l2)
    finally:
        ops.keep_block_activations(mine, on=False)
        ops.clear_grad_views()


def test_bench_config_graph_replay_equals_eager_bf16():
    """bench.py replays the step from a hipGraph: the replayed step must be bit-identical to the eager one (same kernels, same
    order), so the parity shown above for eager launches carries over to the timed path.  Dropout 0.5 as in the benchmark."""
    eeg, spec, labels = (t.to(DEV) for t in _bench_inputs())
    finals = []
    for graphed in (False, True):
        torch.manual_seed(11)
        net = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.5, compute_dtype=torch.bfloat16).to(DEV).train()
        opt = brainxai.FlatAdamW(net.parameters(), lr=1e-3)
        crit = brainxai.KLDivLoss()
        ops.manual_seed(77)
        try:
            if graphed:
                step = brainxai.GraphedTrainStep(net, opt, crit)
                losses = [float(step([eeg, spec], labels)[0]) for _ in range(4)]
                assert step.enabled and len(step._graphs) == 1
            else:
                losses = [float(brainxai.train_step(net, opt, eeg, spec, labels, crit)[0]) for _ in range(4)]
            torch.cuda.synchronize()
            finals.append((losses, opt.flat_p.clone()))
        finally:
            ops.clear_grad_views()
    assert finals[0][0] == finals[1][0], (finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1])


def test_config0_spectrogram_model_alone():
    """configs[0]: Spectrogram_Model (4-plane block1) on 32 synthetic [4,128,256] spectrograms, 6 classes -- the reference's
    CPU-runnable plumbing case (debug_input_size 32, debug_batch_size 16: config.yml:565,569) on the HIP path: full-batch
    forward + backward against the oracle, then the debug epoch (two AdamW steps of B=16, KLDiv batchmean as the unimodal loops
    use, NB:1757) with the first step strict and the epoch's bookkeeping (loss*B accumulation, arg-max accuracy) compared."""
    g = torch.Generator().manual_seed(42)
    spec = torch.rand(32, CIN, H, W, generator=g)
    labels = F.one_hot(torch.randint(0, 6, (32,), generator=g), 6).float()
    ref = O.fill_params(O.Spectrogram_Model(6, in_channels=CIN), seed=23)
    O.set_dropout(ref, 0.0)
    mine = brainxai.Spectrogram_Model(6, in_channels=CIN)
    mine.load_state_dict(ref.state_dict())
    O.set_dropout(mine, 0.0)
    mine.to(DEV)
    ref.train(); mine.train()
    ref0 = copy.deepcopy(ref)
    out_r = ref(spec); loss_r = O.kl_div(out_r, labels, "batchmean"); loss_r.backward()
    try:
        keep = ops.keep_block_activations(mine)
        out = mine(spec.to(DEV)); loss = brainxai.KLDivLoss("batchmean")(out, labels.to(DEV)); loss.backward()
        torch.cuda.synchronize()
        assert rel_err(out.detach().cpu(), out_r.detach()) < TOL and abs(float(loss.detach()) - float(loss_r.detach())) <= TOL * abs(float(loss_r.detach()))
        # gradients against the fp64 oracle with the HIP forward's decisions pinned (the reference's own fp32 gradients are
        # 3e-3 from their fp64 values on this batch: 15 of its ReLU decisions flip -- oracle.ref_torch.conditioning)
        twin, flips = matched_oracle(O, ref0, (spec,), keep, "config0")
        ops.keep_block_activations(mine, on=False)
        O.kl_div(twin(spec.double()), labels.double(), "batchmean").backward()
        fl = _gscale(twin)
        for (n, p), (_, q) in zip(mine.named_parameters(), twin.named_parameters()):
            grad_close(p.grad.cpu(), q.grad, TOL, label=f"config0 d{n}", floor=fl)
        # the debug epoch from the same initial weights
        ref = ref0
        mine.load_state_dict(ref.state_dict())
        opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-3)
        opt_m = brainxai.FlatAdamW(mine.parameters(), lr=1e-3)
        crit = brainxai.KLDivLoss("batchmean")
        tot_r = tot_m = cor_r = cor_m = 0.0
        for k in range(2):
            xs, ys = spec[16 * k:16 * k + 16], labels[16 * k:16 * k + 16]
            opt_r.zero_grad(); o_r = ref(xs); l_r = O.kl_div(o_r, ys, "batchmean"); l_r.backward(); opt_r.step()
            opt_m.zero_grad(); o_m = mine(xs.to(DEV)); l_m = crit(o_m, ys.to(DEV)); l_m.backward(); opt_m.step()
            tot_r += float(l_r) * 16; tot_m += float(l_m) * 16
            cor_r += int((o_r.argmax(1) == ys.argmax(1)).sum()); cor_m += int((o_m.argmax(1).cpu() == ys.argmax(1)).sum())
            if k == 0:
                assert abs(float(l_m) - float(l_r)) <= TOL * abs(float(l_r))
        assert abs(tot_m - tot_r) <= 3e-2 * abs(tot_r) and cor_m == cor_r, (tot_m, tot_r, cor_m, cor_r)
        torch.cuda.synchronize()
        assert int(mine.block3.bn.num_batches_tracked) == int(ref.block3.bn.num_batches_tracked) == 2
    finally:
        ops.keep_block_activations(mine, on=False)
        ops.clear_grad_views()


def test_config4_integrated_gradients_at_full_size():
    """configs[4]: integrated gradients, 50 steps x B=64, zero baselines, at the benchmark's shapes and storage (bf16).
    (a) size-independent property, every sample: completeness -- the attributions of a sample sum to F_c(x) - F_c(baseline), F_c the
        target class's log-probability (50-node Gauss-Legendre quadrature of a piecewise-smooth integrand plus bf16-stored
        activations: observed 3.5 % of the largest |F_c(x) - F_c(0)| of the batch, printed; bound 6 % -- a mis-scaled or missing
        gradient path shows up as tens of percent; the sharp comparison is (c));
    (b) what the sharded sweep computes for a rank's shard equals the rows of the one-shot result (shard_bounds arithmetic on the GPU);
    (c) fp32 storage: the gradients the rule sums (sample 0 at eight of its 50 nodes) strictly (1e-3) against the decision-matched
        fp64 twin; the attributions of two samples against the oracle's fp32 and fp64 integrated gradients (reported; 3e-3)."""
    eeg, spec, _ = _bench_inputs()
    ref, mine = _models(29, torch.bfloat16)
    mine.eval()
    e, s = eeg.to(DEV), spec.to(DEV)
    ie, is_ = brainxai.integrated_gradients(mine, (e, s), None, n_steps=50)
    with torch.no_grad():
        fx = mine(e, s)
        f0 = mine(torch.zeros_like(e), torch.zeros_like(s))
    tgt = fx.argmax(1)
    delta = (fx.gather(1, tgt[:, None]) - f0.gather(1, tgt[:, None]))[:, 0].double().cpu()
    total = (ie.double().flatten(1).sum(1) + is_.double().flatten(1).sum(1)).cpu()
    gap = float((total - delta).abs().max()) / float(delta.abs().max())
    print(f"config4: completeness gap {gap:.3e} of max |dF| {float(delta.abs().max()):.3f}")
    assert gap < 6e-2, gap
    lo, hi = brainxai.shard_bounds(B, 3, 8)                             # rank 3 of 8: samples [24, 32)
    assert (lo, hi) == (24, 32)
    lo_, shard = brainxai.sharded_sweep(lambda a, b: brainxai.integrated_gradients(mine, (a, b), None, n_steps=50)[1], hi - lo, hi - lo,
                                        lambda l, h: (e[lo + l:lo + h], s[lo + l:lo + h]), gather=False)
    assert lo_ == 0 and rel_err(shard.cpu(), is_[lo:hi].cpu()) < 1e-3   # evaluation mode: a sample does not see its batch; only the order of the step sums differs
    ref32, mine32 = _models(29, torch.float32)
    ref32.eval(); mine32.eval()
    # (c) the gradient evaluations the rule sums: sample 0 at eight of the 50 nodes, one batch, strict against the fp64 twin
    from brainxai.explain import _eval_frozen, ig_nodes
    alphas, _ = ig_nodes(50)
    sel = [0, 7, 14, 21, 28, 35, 42, 49]
    xe_k = torch.stack([float(alphas[k]) * eeg[0] for k in sel])
    xs_k = torch.stack([float(alphas[k]) * spec[0] for k in sel])
    with torch.no_grad():
        cls = int(mine32(e[:1], s[:1]).argmax(1))
    onehot = F.one_hot(torch.full((len(sel),), cls), 6).float()
    keep = ops.keep_block_activations(mine32)
    try:
        with _eval_frozen(mine32):
            ek, sk = xe_k.to(DEV).requires_grad_(True), xs_k.to(DEV).requires_grad_(True)
            ge, gs = torch.autograd.grad(mine32(ek, sk), (ek, sk), grad_outputs=onehot.to(DEV))
        torch.cuda.synchronize()
        twin, flips = matched_oracle(O, ref32, (xe_k, xs_k), keep, "config4")
    finally:
        ops.keep_block_activations(mine32, on=False)
    ed, sd = xe_k.double().requires_grad_(True), xs_k.double().requires_grad_(True)
    we, ws_ = torch.autograd.grad(twin(ed, sd), (ed, sd), grad_outputs=onehot.double())
    grad_close(ge.cpu(), we, TOL, label="config4 d eeg at the rule's nodes", flips=flips)
    grad_close(gs.cpu(), ws_, TOL, label="config4 d spec at the rule's nodes", flips=flips)
    # the attribution itself, two samples, against the oracle's own integrated gradients in fp32 and in fp64: unmatched decisions on
    # both sides, so the figures are reported and held to 3e-3; the strict statement is the one above
    je, js = brainxai.integrated_gradients(mine32, (e[:2], s[:2]), None, n_steps=50)
    oe, os_ = O.integrated_gradients(ref32, (eeg[:2], spec[:2]), n_steps=50)
    xe, xs = O.integrated_gradients(copy.deepcopy(ref32).double(), (eeg[:2].double(), spec[:2].double()), n_steps=50)
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    d_hip, d_ref = (l2(je.cpu(), xe), l2(js.cpu(), xs)), (l2(oe, xe), l2(os_, xs))
    print(f"config4: rel-L2 to the fp64 oracle -- HIP fp32: eeg {d_hip[0]:.2e} spec {d_hip[1]:.2e}; oracle fp32: eeg {d_ref[0]:.2e} spec {d_ref[1]:.2e}")
    assert d_hip[0] < TOL and d_hip[1] < 3e-3, (d_hip, d_ref)
