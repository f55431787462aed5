"""CPU: the oracle restatement (oracle/ref_torch.py) replayed against fixtures recorded from the
REFERENCE's own classes (oracle/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as O
from tests.golden_util import GOLDEN, check, load

torch.set_num_threads(min(8, os.cpu_count() or 1))
BLOCKS = {"b4_16_max": (4, 16, 32, 64, "max"), "b16_32_avg": (16, 32, 16, 32, "avg"),
          "b3_16_max_odd": (3, 16, 50, 37, "max"), "b64_128_avg_odd": (64, 128, 25, 18, "avg")}


def test_pin_report_is_clean():
    rep = json.load(open(os.path.join(GOLDEN, "PIN_REPORT.json")))
    worst = max(v for k, v in rep.items() if k != "_meta")
    assert worst < 2e-5


@pytest.mark.parametrize("tag", list(BLOCKS))
def test_block_fwd_bwd(tag):
    cin, c, h, w, kind = BLOCKS[tag]
    fix = load("block_" + tag)
    net = O.fill_params(O.Block(cin, c, kind, (2, 2), dropout_p=0.0), seed=7)
    x = O.seeded((2, cin, h, w), 111 if tag == "b3_16_max_odd" else 11, "randn")
    r = O.seeded((2, c, h // 2, w // 2), 12, "randn")
    check(fix, "x", x, tol=0)
    for mode in ("eval", "train"):
        net.train(mode == "train")
        net.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        (y * r).sum().backward()
        check(fix, f"{mode}.out", y)
        check(fix, f"{mode}.dx", xi.grad)
        for n, p in net.named_parameters():
            check(fix, f"{mode}.grad.{n}", p.grad, tol=2e-5)
    check(fix, "after.running_mean", net.bn.running_mean)
    check(fix, "after.running_var", net.bn.running_var)


@pytest.mark.parametrize("tag,cin", [("spec3_64x96", 3), ("spec4_32x64", 4), ("spec3_100x75", 3)])
def test_spectrogram_model(tag, cin):
    fix = load(tag)
    net = O.fill_params(O.Spectrogram_Model(6, in_channels=cin), seed=21).eval()
    x = torch.from_numpy(fix["x"]) if "x" in fix else None
    if x is None:
        h, w = {"spec3_64x96": (64, 96), "spec3_100x75": (100, 75)}[tag]
        x = O.seeded((2, cin, h, w), 22, "rand")
    check(fix, "eval.logits", net(x))
    check(fix, "eval.block5", net.features(x))
    O.set_dropout(net, 0.0)
    net.train()
    check(fix, "train.logits", net(x))


EEG_CASES = [("eeg19x2000", 19, 2000, {}), ("eeg37x3000", 37, 3000, {}),
             ("eeg_f4d3_70x1024", 70, 1024, dict(F1=4, D=3, F2=8, kernLength=128)),        # outside the default family: the general
             ("eeg_f16d2_5x512", 5, 512, dict(F1=16, D=2, F2=32, kernLength=33))]          # kernel set (csrc/eeg_generic.hip), odd taps too


@pytest.mark.parametrize("tag,chans,samples,kw", EEG_CASES)
def test_eegnet(tag, chans, samples, kw):
    fix = load(tag)
    net = O.fill_params(O.EEGNet(6, Chans=chans, Samples=samples, dropoutRate=0.0, **kw), seed=31)
    x = O.seeded((2, 1, chans, samples), 32, "randn")
    r = torch.from_numpy(fix["r"])
    for mode in ("eval", "train"):
        net.train(mode == "train")
        net.zero_grad()
        xi = x.clone().requires_grad_(True)
        st = net.stages(xi)
        (st["out"] * r).sum().backward()
        check(fix, f"{mode}.out", st["out"])
        check(fix, f"{mode}.conv1.head", st["conv1"][..., :80])
        check(fix, f"{mode}.conv1.tail", st["conv1"][..., -80:])
        for k in ("dw", "bn2", "sep", "bn3"):
            check(fix, f"{mode}.{k}", st[k])
        check(fix, f"{mode}.dx.head", xi.grad[..., :96])
        check(fix, f"{mode}.dx.tail", xi.grad[..., -96:])
        for n, p in net.named_parameters():
            check(fix, f"{mode}.grad.{n}", p.grad, tol=2e-5)
    for k in ("batchnorm1", "batchnorm2", "batchnorm3"):
        check(fix, f"after.{k}.running_var", getattr(net, k).running_var)


@pytest.mark.parametrize("tag,chans,samples,b", [("eegdeep19x2000", 19, 2000, 3), ("eegdeep37x3000", 37, 3000, 2)])
def test_eegnet_attention_deep(tag, chans, samples, b):
    """Row C' (models.py:109-235): third block, attention over time, two dense layers."""
    fix = load(tag)
    net = O.fill_params(O.EEGNetAttentionDeep(6, Chans=chans, Samples=samples, dropoutRate=0.0), seed=61)
    x = O.seeded((b, 1, chans, samples), 62, "randn")
    r = torch.from_numpy(fix["r"])
    for mode in ("eval", "train"):
        net.train(mode == "train")
        net.zero_grad()
        xi = x.clone().requires_grad_(True)
        st = net.stages(xi)
        (st["out"] * r).sum().backward()
        for k in ("out", "conv2", "bn4", "attn", "pool3"):
            check(fix, f"{mode}.{k}", st[k])
        check(fix, f"{mode}.dx.head", xi.grad[..., :96])
        check(fix, f"{mode}.dx.tail", xi.grad[..., -96:])
        for n, p in net.named_parameters():
            check(fix, f"{mode}.grad.{n}", p.grad, tol=2e-5)
    for k in ("batchnorm3", "batchnorm4"):
        check(fix, f"after.{k}.running_var", getattr(net, k).running_var)
    assert int(net.batchnorm4.num_batches_tracked) == int(fix["after.batchnorm4.num_batches_tracked"]) == 1


def test_attention_module():
    """Attention (models.py:109-134) on its own, gradient through both return values."""
    fix = load("attention_32")
    net = O.fill_params(O.Attention(32, 32), seed=71)
    for tag, (b, l) in {"a": (3, 11), "b": (2, 7), "c": (1, 32)}.items():
        x = O.seeded((b, l, 32), 72 + l, "randn").requires_grad_(True)
        r1, r2 = O.seeded((b, l, 32), 73 + l, "randn"), O.seeded((b, l, l), 74 + l, "randn")
        net.zero_grad()
        o, w = net(x)
        ((o * r1).sum() + (w * r2).sum()).backward()
        check(fix, f"{tag}.out", o); check(fix, f"{tag}.weights", w); check(fix, f"{tag}.dx", x.grad)
        fl = 1e-2 * max(float(p.grad.abs().max()) for p in net.parameters())      # key.bias: exactly-zero gradient, rounding noise only
        for n, p in net.named_parameters():
            check(fix, f"{tag}.grad.{n}", p.grad, tol=2e-5, floor=fl)


@pytest.mark.parametrize("tag,cfg", [("mm_bench_small", (19, 2000, 4, 32, 64, 4)),
                                     ("mm_native_small", (37, 3000, 3, 100, 75, 4))])
def test_multimodal_train3(tag, cfg):
    chans, samples, cin, h, w, b = cfg
    fix = load(tag)
    net = O.fill_params(O.build_multimodal(chans, samples, cin, dropout=0.0), seed=41)
    seeds = [int(v) for v in fix["input_seeds"]]
    eeg = O.seeded((b, 1, chans, samples), seeds[0], "randn")
    spec = O.seeded((b, cin, h, w), seeds[1], "rand")
    assert float(fix["conditioning"][0]) < 1e-4          # the recorded fp32 run is a well-posed target (make_golden checks flips too)
    labels = torch.from_numpy(fix["labels"])
    net.eval()
    y = net(eeg, spec)
    check(fix, "eval.logits", y)
    check(fix, "eval.loss_mean", O.kl_div(y, labels, "mean"))
    check(fix, "eval.loss_batchmean", O.kl_div(y, labels, "batchmean"))
    onehot = torch.nn.functional.one_hot(labels.argmax(1), 6).float()
    check(fix, "eval.loss_onehot", O.kl_div(y, onehot, "mean"))
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    losses = []
    for step in range(3):
        loss, _ = O.train_step(net, opt, eeg, spec, labels)
        losses.append(loss)
        if step == 0:
            for n, p in net.named_parameters():
                check(fix, "step0.ghead." + n, p.grad.flatten()[:32], tol=2e-5)
    check(fix, "train.losses", np.array(losses))
    for n, t in net.state_dict().items():
        check(fix, "after3.shead." + n, t.float().flatten()[:32], tol=2e-5)


def test_ddp_microbatch_gradients_and_mean():
    """SURVEY 8(c) fixture 9: gradients of the reference classes on 8 micro-batches and their mean (what all-reduce(AVG) of the
    flat gradient arena must give): the oracle reproduces each rank's gradient and the mean"""
    fix = load("ddp8_bench_small")
    world, b = int(fix["world"][0]), int(fix["batch"][0])
    mean = None
    for r in range(world):
        net = O.fill_params(O.build_multimodal(19, 2000, 4, dropout=0.0), seed=41).train()
        eeg, spec = O.seeded((b, 1, 19, 2000), 420 + r, "randn"), O.seeded((b, 4, 32, 64), 430 + r, "rand")
        labels = torch.softmax(O.seeded((b, 6), 440 + r, "randn"), 1)
        O.kl_div(net(eeg, spec), labels).backward()
        g = torch.cat([p.grad.flatten() for p in net.parameters()])
        check(fix, f"rank{r}.ghead", g[:64], tol=2e-5)
        np.testing.assert_allclose(O.summarize(g)[[1, 3]], fix[f"rank{r}.gsum"][[1, 3]], rtol=1e-5)
        mean = g.double() if mean is None else mean + g.double()
    mean = (mean / world).float()
    np.testing.assert_allclose(O.summarize(mean)[[1, 3]], fix["mean.gsum"][[1, 3]], rtol=1e-5)
    off = 0
    for n, p in net.named_parameters():
        check(fix, "mean.ghead." + n, mean[off:off + min(32, p.numel())], tol=2e-5, floor=1e-2 * float(fix["mean.gmax"][0]))
        off += p.numel()


def test_gradcam_saliency_ig():
    net = O.fill_params(O.build_multimodal(19, 2000, 4, dropout=0.0), seed=51).eval()
    eeg = O.seeded((2, 1, 19, 2000), 52, "randn")
    fix = load("gradcam_4x64x128")
    spec = O.seeded((2, 4, 64, 128), int(fix["spec_seed"][0]), "rand")
    for layer in ("block5", "block5.conv3", "block3"):
        cam, raw, w, A, out = O.grad_cam(net, eeg, spec, "spectrogram_model." + layer, "all",
                                         upsample=False, return_parts=True)
        check(fix, layer + ".raw", raw); check(fix, layer + ".cam", cam); check(fix, layer + ".w", w)
    check(fix, "up.block5", O.grad_cam(net, eeg, spec, class_idx="all"))
    check(fix, "argmax.block5", O.grad_cam(net, eeg, spec))
    sal = load("saliency_4x64x128")
    se, ss = O.saliency(net, eeg[:1], spec[:1], reference_quirk=True)
    check(sal, "eeg_ref", se[0]); check(sal, "spec_ref_x2", ss[0])
    ig = load("ig_4x32x64")
    ie, is_ = O.integrated_gradients(net, (eeg[:1], spec[:1, :, :32, :64].contiguous()), n_steps=50)
    check(ig, "eeg_attr", ie, tol=5e-5); check(ig, "spec_attr", is_, tol=5e-5)


def test_stacker():
    fix = load("stacker_2x10000x19")
    raw = O.synthetic_batch(batch=2, seed=61, stacked=False)["raw_eeg"].numpy()
    out = np.stack([O.eeg_transform(r) for r in raw])
    check(fix, "out", out)
    b, a = O.butter_lowpass_coeffs()
    np.testing.assert_allclose(b, fix["b"], rtol=1e-12)
    np.testing.assert_allclose(a, fix["a"], rtol=1e-12)
    assert O.stack_eeg_batch(raw).shape == (2, 1, 19, 2000)


def test_state_dict_manifest():
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))
    nets = {"Block(4,16)": O.Block(4, 16), "Spectrogram_Model": O.Spectrogram_Model(6),
            "EEGNet(6,19,2000)": O.EEGNet(6, Chans=19, Samples=2000), "EEGNet(6,37,3000)": O.EEGNet(6),
            "EEGNetAttentionDeep(6,19,2000)": O.EEGNetAttentionDeep(6, Chans=19, Samples=2000),
            "EEGNetAttentionDeep(6,37,3000)": O.EEGNetAttentionDeep(6),
            "EEGNetAttentionDeep(6,19,2000)": O.EEGNetAttentionDeep(6, Chans=19, Samples=2000),
            "EEGNetAttentionDeep(6,37,3000)": O.EEGNetAttentionDeep(6),
            "MultimodalModel(bench)": O.build_multimodal(19, 2000, 4),
            "MultimodalModel(native)": O.build_multimodal(37, 3000, 3)}
    for name, net in nets.items():
        assert {k: list(v.shape) for k, v in net.state_dict().items()} == man[name], name
        assert sum(p.numel() for p in net.parameters()) == man[name + "#params"]
    assert man["MultimodalModel(bench)#params"] == 2025074


def test_montage_stacker():
    """8(f) rank 3: oracle restatement of CombinedDataset.process_eeg against vectors made by the reference's own methods"""
    fix = load("montage_2x10000x20")
    frames = O.synthetic_frames(batch=2, seed=7)
    assert int(np.isnan(frames).sum()) == int(fix["nan_count"][0])
    got = np.stack([O.montage_transform(f) for f in frames])
    for r in (0, 7, 18, 19, 20, 28, 36):
        check(fix, f"row{r}", got[:, 0, r, :2560], tol=1e-6)
    check(fix, "full", got, tol=1e-6)
    rows = O.montage_rows()
    assert len(rows) == 37 and rows[19] == (19, -1) and rows[20] == (0, 4) and rows[36] == (8, 9)


def test_spectrogram_preprocessing():
    """8(f) rank 2: oracle restatement of CombinedDataset.process_spectrogram against vectors made by the reference's own methods"""
    fix = load("specprep_2x320x400")
    frames = O.synthetic_spectrogram_frames(batch=2, seed=5)
    assert int(np.isnan(frames).sum()) == int(fix["nan_count"][0])
    got = np.stack([O.spectrogram_transform(f.astype(np.float64), off) for f, off in zip(frames, (None, 60))])
    check(fix, "plane", got[:, 0, ::8, ::6], tol=1e-6)
    check(fix, "full", got, tol=1e-6)
    w = O.gaussian_weights()
    assert len(w) == 9 and abs(w.sum() - 1) < 1e-15 and w[4] == w.max()


def test_distributed_epoch_loop_is_pinned_by_the_reference_function():
    """oracle.distributed_epoch (row F's loop restatement) replays the histories that the reference's OWN
    train_and_validate_eeg_distributed produced in the build container (oracle/make_golden.py gen_ddp_loop: the function is
    ast-extracted from training_distributed.py:22-141 and run with dist / DDP / checkpoint I/O stubbed)."""
    import torch.nn as nn
    fix = load("ddp_loop_eeg_19x2000")
    wd, lr0 = float(fix["weight_decay"][0]), float(fix["lr0"][0])
    net = O.fill_params(O.EEGNet(6, Chans=19, Samples=2000, dropoutRate=0.0), seed=91)
    batches = lambda seed0, n: [(O.seeded((4, 1, 19, 2000), seed0 + i, "randn"), torch.softmax(O.seeded((4, 6), seed0 + 50 + i, "randn"), 1)) for i in range(n)]   # noqa: E731
    train, valid = batches(700, 3), batches(800, 2)
    opt = torch.optim.AdamW(net.parameters(), lr=lr0)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=0, threshold=10.0)
    hist = [O.distributed_epoch(net, train, valid, opt, nn.KLDivLoss(), wd, sched) for _ in range(2)]
    for key, name in (("train_loss", "train_losses"), ("reg_loss", "regularization_losses"), ("valid_loss", "valid_losses"),
                      ("train_acc", "train_accuracies"), ("valid_acc", "valid_accuracies"), ("lr", "lr_scheduler")):
        np.testing.assert_allclose([h[key] for h in hist], fix[name], rtol=1e-6, err_msg=name)
    for n, t in net.state_dict().items():
        check(fix, "final.shead." + n, t.detach().float().flatten()[:32], tol=1e-5)
