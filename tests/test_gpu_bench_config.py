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


def test_overlapped_graph_walks_the_serial_trajectory():
    """round 3: inside a captured step the EEG branch runs on a side stream beside the spectrogram branch (ops.OVERLAP_EEG = 1, the
    default).  Same kernels, same arithmetic, so the replayed step must walk the SERIAL eager trajectory bit for bit -- 30 steps at the
    benchmark's shape, dropout on -- and 100 replays of the captured forward must reproduce the serial forward's branch outputs.
    (This is the test that found hipcc's packed-fp32 code misbehaving beside another kernel: build.py compiles without it.)"""
    assert ops.OVERLAP_EEG == 1, "the default is what bench.py measures"
    eeg, spec, labels = (t.to(DEV) for t in _bench_inputs())
    finals = []
    for graphed in (False, True):
        torch.manual_seed(11)
        net = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.5, compute_dtype=torch.bfloat16).to(DEV).train()
        opt = brainxai.FlatAdamW(net.parameters(), lr=1e-3)
        crit = brainxai.KLDivLoss()
        ops.manual_seed(78)
        try:
            if graphed:
                step = brainxai.GraphedTrainStep(net, opt, crit)
                losses = [float(step([eeg, spec], labels)[0]) for _ in range(30)]     # (the graph's loss is one static buffer: read it per step)
                assert step.enabled and len(step._graphs) == 1
            else:
                losses = [float(brainxai.train_step(net, opt, eeg, spec, labels, crit)[0]) for _ in range(30)]
            torch.cuda.synchronize()
            finals.append((losses, opt.flat_p.clone()))
        finally:
            ops.clear_grad_views()
    assert finals[0][0] == finals[1][0], [(i, a, b) for i, (a, b) in enumerate(zip(*[f[0] for f in finals])) if a != b][:3]
    assert torch.equal(finals[0][1], finals[1][1])
    # the forward alone, replayed: both branches' outputs against the serial launch order
    sm, em = net.spectrogram_model, net.eeg_model

    def fwd():
        with torch.no_grad():
            xi, ss, se = sm._pack_all(spec, seed_pair=True)
            if ops.overlap_eeg_now():
                cur, side = ops.fork_eeg(DEV, se, eeg)
                with torch.cuda.stream(side):
                    ef = em.features(eeg, seed=se)
                sf = sm.features(spec, seed=ss, packed=(xi,))
                cur.wait_stream(side)
                ef.record_stream(cur)
                return ef, sf
            return em.features(eeg, seed=se), sm.features(spec, seed=ss, packed=(xi,))
    ops.manual_seed(1234)
    ref = [t.clone() for t in fwd()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fwd()
    bad = 0
    for _ in range(100):
        ops.manual_seed(1234)
        graph.replay()
        torch.cuda.synchronize()
        bad += int(not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])))
    assert bad == 0, f"{bad} of 100 replays of the overlapped forward differ from the serial forward"


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
    """configs[4]: integrated gradients, 50 steps x B=64, zero baselines, at the benchmark's shapes.
    (a) completeness, every sample, bf16 storage (the benchmarked dtype) and fp32 storage: the attributions of a sample sum to
        F_c(x) - F_c(baseline), F_c the target class's log-probability, up to the error of the 50-node Gauss-Legendre rule on a
        piecewise-smooth integrand.  That quadrature error is MEASURED, not guessed: part (c) evaluates the same rule in fp64
        (decision-matched twin) for two samples -- gaps 0.120 and 0.035 of |dF| = 5.48, i.e. 2.2e-2 and 6e-3 -- and the fp32 path
        must reproduce the twin's gap to 1e-3 of |dF| on those samples (observed 1e-4).  Over the 64 samples the largest gap is
        4.2e-2 (fp32) / 3.1e-2 (bf16) of the batch's largest |dF|: the rule's error, of either sign, not the storage type's; both
        are held to 6e-2 (a mis-scaled or missing gradient path shows up as tens of percent).
    (b) what the sharded sweep computes for a rank's shard equals the rows of the one-shot result (shard_bounds arithmetic on the GPU);
    (c) fp32 storage, the WHOLE attribution of two samples: the gradient at every one of the 50 nodes strictly (1e-3) against the
        fp64 twin whose ReLU / max-pool decisions are pinned to that very forward's (every disagreement a demonstrated tie), the
        twin's gradients summed with the rule's weights, and ``integrated_gradients``' output held to 1e-3 of that (round 2 compared
        the attribution with the oracle's own fp32 / fp64 runs -- unmatched decisions on both sides -- and had to allow 3e-3;
        those figures are still printed)."""
    eeg, spec, _ = _bench_inputs()
    ref, mine = _models(29, torch.bfloat16)
    mine.eval()
    e, s = eeg.to(DEV), spec.to(DEV)
    ref32, mine32 = _models(29, torch.float32)
    ref32.eval(); mine32.eval()

    def completeness(model):
        ie_, is__ = brainxai.integrated_gradients(model, (e, s), None, n_steps=50)
        with torch.no_grad():
            fx = model(e, s)
            f0 = model(torch.zeros_like(e), torch.zeros_like(s))
        tgt = fx.argmax(1)
        delta = (fx.gather(1, tgt[:, None]) - f0.gather(1, tgt[:, None]))[:, 0].double().cpu()
        total = (ie_.double().flatten(1).sum(1) + is__.double().flatten(1).sum(1)).cpu()
        return ie_, is__, total - delta, delta, tgt
    ie, is_, gap16, delta16, tgt16 = completeness(mine)
    je_all, js_all, gap32, delta32, tgt32 = completeness(mine32)
    dmax = float(delta32.abs().max())
    same = (tgt16 == tgt32).cpu()                                        # a near-tie of two classes can pick another target in bf16
    print(f"config4: completeness gap / max |dF| ({dmax:.3f}): bf16 {float(gap16.abs().max()) / dmax:.3e}, fp32 {float(gap32.abs().max()) / dmax:.3e}; "
          f"{int(same.sum())} of {B} samples share the target class, largest bf16 - fp32 gap difference among them {float((gap16 - gap32)[same].abs().max()) / dmax:.3e}")
    assert int(same.sum()) >= B - 4
    assert float(gap16.abs().max()) < 6e-2 * dmax and float(gap32.abs().max()) < 6e-2 * dmax
    lo, hi = brainxai.shard_bounds(B, 3, 8)                             # rank 3 of 8: samples [24, 32)
    assert (lo, hi) == (24, 32)
    lo_, shard = brainxai.sharded_sweep(lambda a, b: brainxai.integrated_gradients(mine, (a, b), None, n_steps=50)[1], hi - lo, hi - lo,
                                        lambda l, h: (e[lo + l:lo + h], s[lo + l:lo + h]), gather=False)
    assert lo_ == 0 and rel_err(shard.cpu(), is_[lo:hi].cpu()) < 1e-3   # evaluation mode: a sample does not see its batch; only the order of the step sums differs
    # (c) every gradient evaluation the rule sums for samples 0 and 1, strict against the decision-matched fp64 twin, ten nodes at a time
    from brainxai.explain import _eval_frozen, ig_nodes
    alphas, wts = ig_nodes(50)
    NS, CH = 2, 10
    tw_e, tw_s = torch.zeros(NS, *eeg.shape[1:], dtype=torch.float64), torch.zeros(NS, *spec.shape[1:], dtype=torch.float64)
    nflips = 0
    for i in range(NS):
        cls = int(tgt32[i])
        for c0 in range(0, 50, CH):
            ks = list(range(c0, c0 + CH))
            xe_k = torch.stack([float(alphas[k]) * eeg[i] for k in ks])
            xs_k = torch.stack([float(alphas[k]) * spec[i] for k in ks])
            onehot = F.one_hot(torch.full((CH,), cls), 6).float()
            keep = ops.keep_block_activations(mine32)
            try:
                with _eval_frozen(mine32):
                    ek, sk = xe_k.to(DEV).requires_grad_(True), xs_k.to(DEV).requires_grad_(True)
                    ge, gs = torch.autograd.grad(mine32(ek, sk), (ek, sk), grad_outputs=onehot.to(DEV))
                torch.cuda.synchronize()
                twin, flips = matched_oracle(O, ref32, (xe_k, xs_k), keep, f"config4 sample {i} nodes {c0}..{c0 + CH - 1}")
            finally:
                ops.keep_block_activations(mine32, on=False)
            nflips += len(flips)
            ed, sd = xe_k.double().requires_grad_(True), xs_k.double().requires_grad_(True)
            we, ws_ = torch.autograd.grad(twin(ed, sd), (ed, sd), grad_outputs=onehot.double())
            grad_close(ge.cpu(), we, TOL, label=f"config4 d eeg, sample {i}, nodes {c0}..", flips=flips)
            grad_close(gs.cpu(), ws_, TOL, label=f"config4 d spec, sample {i}, nodes {c0}..", flips=flips)
            for j, k in enumerate(ks):
                tw_e[i] += float(wts[k]) * we[j]
                tw_s[i] += float(wts[k]) * ws_[j]
    tw_e, tw_s = tw_e * eeg[:NS].double(), tw_s * spec[:NS].double()          # zero baselines: (x - x') = x
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    mx = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    je, js = je_all[:NS].cpu(), js_all[:NS].cpu()
    d_twin = (l2(je, tw_e), l2(js, tw_s), mx(je, tw_e), mx(js, tw_s))
    print(f"config4: integrated_gradients (fp32 storage) vs the decision-matched fp64 attribution, {NS} samples x 50 nodes, {nflips} pinned ties: "
          f"rel-L2 eeg {d_twin[0]:.2e} spec {d_twin[1]:.2e}; max-abs on the largest entry eeg {d_twin[2]:.2e} spec {d_twin[3]:.2e}")
    assert max(d_twin) < TOL, d_twin
    # the rule's own error (completeness gap of the fp64 twin attribution) is what the fp32 path must show on these samples
    gap_twin = (tw_e.flatten(1).sum(1) + tw_s.flatten(1).sum(1)) - delta32[:NS]
    print(f"config4: quadrature gap of the 50-node rule in fp64 (samples 0..{NS - 1}): {[f'{float(v):.4f}' for v in gap_twin]}; fp32 path: "
          f"{[f'{float(v):.4f}' for v in gap32[:NS]]}  (|dF| {[f'{float(v):.3f}' for v in delta32[:NS]]})")
    assert float((gap_twin - gap32[:NS]).abs().max()) < TOL * dmax
    # for the record: the same attribution against the oracle's OWN fp32 and fp64 runs (unmatched decisions on both sides)
    oe, os_ = O.integrated_gradients(ref32, (eeg[:2], spec[:2]), n_steps=50)
    xe, xs = O.integrated_gradients(copy.deepcopy(ref32).double(), (eeg[:2].double(), spec[:2].double()), n_steps=50)
    d_hip, d_ref = (l2(je, xe), l2(js, xs)), (l2(oe, xe), l2(os_, xs))
    print(f"config4: rel-L2 to the fp64 oracle -- HIP fp32: eeg {d_hip[0]:.2e} spec {d_hip[1]:.2e}; oracle fp32: eeg {d_ref[0]:.2e} spec {d_ref[1]:.2e}")
    assert d_hip[0] < TOL and d_hip[1] < 2e-3, (d_hip, d_ref)


def test_bf16_gradcam_sweep_against_fp32_oracle():
    """The maps the benchmarked Grad-CAM sweep (configs[3]) produces -- bf16 storage, exactly GradCamSweep's launches: pair kernels
    without the conv1 store, collapsed EEG evaluation path, bx_gradcam_head_sweep -- against ``O.grad_cam`` on the fp32 oracle, and
    the fp32-storage sweep beside it (every Grad-CAM parity test ran fp32 storage, the benchmarked maps' error was
    unquantified).  16 samples x 6 classes at the benchmark's [4,128,256] / [1,19,2000].

    Derived bound.  A map is raw[p] = sum_c w_c A[p,c] over the 256 channels of block5's output A.  bf16 storage rounds every stored
    activation of the 15 convolution layers and 5 stage outputs to 8 significand bits (2^-9 relative, round to nearest): A arrives
    with a relative error of ~sqrt(20) 2^-9 = 9e-3 of its scale at worst, 4e-3 observed (asserted: 2^-7).  The channel sum CANCELS:
    at random initialisation max_p sum_c |w_c A_c| is ~40 x max |raw|, so an error of eps in A is an error of up to
    eps * max_p sum_c |w_c A_c| in the map -- on the scale of the largest pre-ReLU map value that is eps x the cancellation ratio
    R (computed here from the oracle's own parts).  Asserted: err <= 2^-7 R (the worst case, errors of one sign) and, as the
    regression guard, err <= 2^-7 sqrt(R) x 3 (errors of random sign add in quadrature: observed 1.8e-2 at R = 38).
    north_star's 1e-3 belongs to fp32 storage: asserted for the fp32 sweep (observed 3e-6)."""
    g = torch.Generator().manual_seed(4242)
    n = 16
    spec = torch.rand(n, CIN, H, W, generator=g)
    eeg = torch.randn(n, 1, CHANS, T, generator=g)
    ref = O.fill_params(O.build_multimodal(CHANS, T, CIN, dropout=0.0), seed=5).eval()
    cam_o, raw_o, w_o, A_o, out_o = O.grad_cam(ref, eeg, spec, "spectrogram_model.block5", "all", upsample=False, return_parts=True)
    up_o = O.grad_cam(ref, eeg, spec, class_idx="all")                                   # [16, 6, 128, 256]
    scale = float(raw_o.abs().max())
    ratio = float((w_o.abs()[:, :, :, None, None] * A_o.abs()[:, None]).sum(2).max()) / scale
    errs = {}
    for dt in (torch.float32, torch.bfloat16):
        mine = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.0, compute_dtype=dt)
        mine.load_state_dict(ref.state_dict())
        mine.to(DEV).eval()
        e, s = eeg.to(DEV), spec.to(DEV)
        small = brainxai.GradCamSweep(mine, e, s, class_idx="all", upsample=False)(e, s).float().cpu()
        up = brainxai.GradCamSweep(mine, e, s, class_idx="all")(e, s).float().cpu()
        assert small.shape == cam_o.shape and up.shape == up_o.shape == (n, 6, H, W)
        _, _, _, A, _ = brainxai.grad_cam(mine, e, s, "spectrogram_model.block5", "all", upsample=False, return_parts=True)
        A = A.float().cpu()
        if A.shape != A_o.shape:
            A = A.permute(0, 3, 1, 2)
        errs[dt] = (float((small.double() - cam_o.double()).abs().max()) / scale, float((up.double() - up_o.double()).abs().max()) / scale,
                    float((A.double() - A_o.double()).abs().max() / A_o.double().abs().max()))
    torch.cuda.synchronize()
    print(f"Grad-CAM sweep vs fp32 oracle on the raw-map scale (cancellation ratio R = {ratio:.1f}): "
          f"bf16 maps {errs[torch.bfloat16][0]:.2e} (upsampled {errs[torch.bfloat16][1]:.2e}, block5 output {errs[torch.bfloat16][2]:.2e}); "
          f"fp32 maps {errs[torch.float32][0]:.2e} (upsampled {errs[torch.float32][1]:.2e}, block5 output {errs[torch.float32][2]:.2e})")
    assert max(errs[torch.float32][:2]) < TOL and errs[torch.float32][2] < 1e-5
    eps = 2.0 ** -7
    assert errs[torch.bfloat16][2] < eps
    assert max(errs[torch.bfloat16][:2]) < eps * ratio
    assert max(errs[torch.bfloat16][:2]) < 3 * eps * ratio ** 0.5
