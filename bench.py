#!/usr/bin/env python3
"""Headline benchmark: multimodal (SpectrogramCNN + EEGNet + fusion) TRAINING samples/sec on synthetic
[64,4,128,256] spectrograms + [64,10000,19] raw EEG per GPU (BASELINE.json configs[1]; configs[2] under
--gpus 8), with Grad-CAM maps/sec, the dominant kernel's roofline and the CPU oracle timed beside it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero_grad -> forward -> KLDiv -> backward -> (RCCL all-reduce of the flat gradient arena) -> fused
AdamW over one B=64 batch per rank, inputs resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
B, CIN, H, W, CHANS, RAW_LEN, T = 64, 4, 128, 256, 19, 10000, 2000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gradcam", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    return ap.parse_args()


def conv_work(bx, dtype_bytes):
    """Algorithmic FLOPs / bytes of every conv3x3 launch of one training step:
    forward 15 launches, data-gradient 10 (block1.conv1's is skipped: the input needs no gradient... but
    conv1 of blocks 2-5 do), weight-gradient 15.  Returns dict kind -> (flops, bytes, launches)."""
    stages = [(8, 16, 128, 256), (16, 32, 64, 128), (32, 64, 32, 64), (64, 128, 16, 32), (128, 256, 8, 16)]
    out = {"fwd": [0.0, 0.0, 0], "dgrad": [0.0, 0.0, 0], "wgrad": [0.0, 0.0, 0]}
    for si, (cin_p, c, h, w) in enumerate(stages):
        px = bx * h * w
        for k, ci in enumerate((cin_p, c, c)):
            fl = 2.0 * 9 * ci * c * px
            by = px * (ci + c) * dtype_bytes
            out["fwd"][0] += fl; out["fwd"][1] += by; out["fwd"][2] += 1
            out["wgrad"][0] += fl; out["wgrad"][1] += by; out["wgrad"][2] += 1
            if not (si == 0 and k == 0):
                extra = px * (ci if k else 0) * dtype_bytes      # ReLU-mask read (conv2/3) or skip addend (conv1)
                out["dgrad"][0] += fl; out["dgrad"][1] += by + px * ci * dtype_bytes; out["dgrad"][2] += 1
                del extra
    return out


def pmc_conv_traffic(dtype):
    """HBM bytes per conv3x3-family launch from the newest committed PMC summary (profiles/*_pmc_hbm_traffic.txt written by
    tools/step_profile.py --pmc: same model, batch and dtype as this benchmark); None when absent or for another dtype."""
    import glob
    if dtype != "bf16":
        return None
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.txt")))
    for path in reversed(files):
        tot, launches = 0.0, 0.0
        for line in open(path):
            if not line.startswith(("k_conv_mfma", "k_conv3x3", "k_wgrad_mfma", "k_wgrad_reduce")):
                continue
            try:
                wr, fe, n = float(line.split()[-1]), float(line.split()[-2]), float(line.split()[-3])
            except ValueError:
                continue
            tot += n * (fe + wr) * 1e6
            if not line.startswith("k_wgrad_reduce"):
                launches += n
        if launches > 0:
            return round(tot / launches)
    return None


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: library banners (RCCL prints its version to stdout at init) and any
    # other chatter are routed to stderr for the whole run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("BX_BENCH_FORCE_DIST") == "1"      # exercise the data-parallel path with a 1-rank RCCL group
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import brainxai
    from brainxai import ops
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    # ---- synthetic inputs, seed 42 + rank, generated on the host like a DataLoader would
    g = torch.Generator().manual_seed(42 + rank)
    spec = torch.rand(B, CIN, H, W, generator=g).to(dev)
    raw = (torch.randn(B, RAW_LEN, CHANS, generator=g) * 100.0)
    flat = raw.view(-1)
    k = flat.numel() // 1000
    idx = torch.randint(0, flat.numel(), (2 * k,), generator=g)
    flat[idx[:k]] = float("nan"); flat[idx[k:]] *= 50.0
    raw = raw.to(dev)
    labels = torch.softmax(torch.randn(B, 6, generator=g), 1).to(dev)
    eeg = brainxai.stack_eeg(raw)                      # [B,1,19,2000], resident
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        brainxai.stack_eeg(raw)
    torch.cuda.synchronize()
    stacker_sps = 3 * B / (time.perf_counter() - t0)
    # the notebook's native montage chain, raw frames [B,10000,20] -> [B,1,37,3000]
    frames = torch.randn(B, 10000, 20, device=dev) * 50
    brainxai.stack_eeg_montage(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        brainxai.stack_eeg_montage(frames)
    torch.cuda.synchronize()
    montage_sps = 3 * B / (time.perf_counter() - t0)
    del frames
    # the notebook's native spectrogram chain, parquet values [B,320,400] -> [B,3,400,300]
    sframes = torch.rand(B, 320, 400, device=dev) * 40
    brainxai.preprocess_spectrograms(sframes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        brainxai.preprocess_spectrograms(sframes)
    torch.cuda.synchronize()
    specprep_sps = 3 * B / (time.perf_counter() - t0)
    del sframes

    torch.manual_seed(42)
    model = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.5, compute_dtype=cdt).to(dev).train()
    ddp = brainxai.DataParallel(model) if (world > 1 or force_dist) else None
    opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
    crit = brainxai.KLDivLoss()

    def fwd_bwd():
        opt.zero_grad()
        out = model(eeg, spec)
        loss = crit(out, labels)
        loss.backward(ops.unit_gradient(loss.device))       # same as loss.backward(): the seed gradient 1.0 is a cached tensor, not a fill launch
        return loss.detach()

    def finish():
        if ddp is not None:
            ddp.sync_gradients(opt)           # ONE RCCL all-reduce(AVG) of the flat gradient arena
            opt.step(gathered=True)
        else:
            opt.step()

    def step():
        loss = fwd_bwd()
        finish()
        return loss, None

    for _ in range(args.warmup):
        loss, _ = step()
    # ---- hipGraph: ~130 launches per step are captured once and replayed.  Single GPU: the whole step.  Multi GPU:
    # forward+backward are replayed, the all-reduce and the fused AdamW (2 launches) are issued eagerly after it.
    graph, graph_covers_all = None, ddp is None
    if not args.no_graph:
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                    # allocate everything once on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g_loss = fwd_bwd()
                if graph_covers_all:
                    finish()
            graph.replay()
            if not graph_covers_all:
                finish()
            torch.cuda.synchronize()
        except Exception as exc:                          # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            graph = None
    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
            if not graph_covers_all:
                finish()
        else:
            loss, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if graph is not None:
        loss = g_loss
    # ---- per-kernel HIP-event timing of the conv family: the same step launched eagerly right after the timed
    # region (events cannot bracket kernels inside a replayed graph), same buffers, same data
    prof = []
    ops.CONV_PROFILE = prof
    prof_steps = min(args.steps, 5)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    ops.CONV_PROFILE = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    loss_val = float(loss)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- dominant kernel (conv3x3 family): HIP-event durations recorded around each launch in the timed region
    kinds = {}
    for kind, ev0, ev1 in prof:
        kinds.setdefault(kind, []).append(ev0.elapsed_time(ev1) * 1e-3)
    work = conv_work(B, 2 if args.dtype == "bf16" else 4)
    conv_time = sum(sum(v) for v in kinds.values()) / prof_steps            # seconds per step in conv kernels
    conv_flops = sum(w[0] for w in work.values())
    conv_bytes = sum(w[1] for w in work.values())
    n_launch = sum(len(v) for v in kinds.values())
    roofline = None
    extra = {}
    if conv_time > 0:
        ach_gbs = conv_bytes / conv_time / 1e9
        ach_tf = conv_flops / conv_time / 1e12
        roofline = {"bound": "mfma", "kernel": "conv3x3 fwd+dgrad+wgrad (%d launches/step)" % (n_launch // max(prof_steps, 1)),
                    "achieved": round(ach_tf, 3), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach_tf / MFMA_PEAK_TFLOPS[args.dtype], 5), "traffic": pmc_conv_traffic(args.dtype),
                    "avg_launch_us": round(conv_time / max(n_launch / prof_steps, 1) * 1e6, 2),
                    "share_of_step": round(conv_time / (elapsed / args.steps), 3)}
        if roofline["traffic"] is not None:
            roofline["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, conv3x3 family average; committed summary, not live)"
            roofline["algorithmic_bytes_per_launch"] = round(conv_bytes / max(n_launch / prof_steps, 1))
        extra["roofline_hbm"] = {"bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ach_gbs / HBM_PEAK_GBS, 5)}
        extra["conv_ms_per_step"] = {k_: round(sum(v) / prof_steps * 1e3, 3) for k_, v in kinds.items()}
    # whole-step algorithmic roofline (5.71 GFLOP and 52.2 MB bf16 / 104.5 MB fp32 per sample)
    per_sample_bytes = 52.2e6 if args.dtype == "bf16" else 104.5e6
    extra["step_roofline"] = {"hbm_frac": round(per_sample_bytes * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                              "mfma_frac": round(5.71e9 * B / (elapsed / args.steps) / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5)}

    # ---- Grad-CAM maps/sec (configs[3] shape: eval, target block5, all 6 classes, upsampled to 128x256)
    gradcam = None
    if not args.no_gradcam:
        model.eval()
        for _ in range(2):
            brainxai.grad_cam(model, eeg, spec, class_idx="all")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            maps = brainxai.grad_cam(model, eeg, spec, class_idx="all")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        eager_maps = world * reps * maps.shape[0] * maps.shape[1] / dt
        # the sweep form: the same launches replayed from a hipGraph (what a 10 000-sample sweep uses)
        sweep = brainxai.GradCamSweep(model, eeg, spec, class_idx="all")
        sweep(eeg, spec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            maps = sweep(eeg, spec)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        gradcam = {"maps_per_sec": round(world * reps * maps.shape[0] * maps.shape[1] / dt, 1),
                   "samples_per_sec": round(world * reps * B / dt, 1), "classes": int(maps.shape[1]), "target": "spectrogram_model.block5",
                   "mode": "hipGraph sweep (GradCamSweep); eager grad_cam() calls: %.0f maps/s" % eager_maps}
        del sweep
        # configs[4]-style integrated gradients (n_steps=50, zero baselines) on 8 samples of the batch: 50 fwd + dgrad sweeps
        ig_in = (eeg[:8], spec[:8])
        brainxai.integrated_gradients(model, ig_in, None, n_steps=50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        brainxai.integrated_gradients(model, ig_in, None, n_steps=50)
        torch.cuda.synchronize()
        gradcam["ig50_samples_per_sec"] = round(8 / (time.perf_counter() - t0), 2)
        model.train()

    # ---- measured device-to-device copy rate (quote the box's own streaming rate next to the 8 TB/s spec)
    if rank == 0:
        src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        extra["measured_copy_GBps"] = round(10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)   # read + write bytes
        del src, dst

    # ---- CPU baseline: the oracle (a port of the reference's PyTorch path) on this box's host cores, rank 0, N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_torch as O
        # the GPU box shows every host CPU (os.cpu_count() = 256) but grants one GPU a 16-CPU share:
        # more threads than that only oversubscribes (measured 0.99 samples/s at 256 threads)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(avail, 16)))
        ref = O.build_multimodal(CHANS, T, CIN, dropout=0.5).train()
        ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
        ce, cs, cl = eeg.cpu(), spec.cpu(), labels.cpu()
        O.train_step(ref, ropt, ce, cs, cl)
        t0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            O.train_step(ref, ropt, ce, cs, cl)
        cdt_s = time.perf_counter() - t0
        cpu = {"value": round(args.cpu_steps * B / cdt_s, 2), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{args.cpu_steps} training steps of the same B={B} batch (oracle/ref_torch.py, fp32, "
                         f"{torch.get_num_threads()} torch threads; host shows {os.cpu_count()} CPUs, share is 16)"}
        if gradcam is not None:
            ref.eval()
            t0 = time.perf_counter()
            O.grad_cam(ref, ce[:16], cs[:16], class_idx="all")
            cpu["gradcam_maps_per_sec"] = round(16 * 6 / (time.perf_counter() - t0), 2)
        fr = O.synthetic_frames(batch=4, seed=3)
        t0 = time.perf_counter()
        for f in fr:
            O.montage_transform(f)
        cpu["montage_stacker_samples_per_sec"] = round(4 / (time.perf_counter() - t0), 2)     # one host core, numpy/scipy
        sf = O.synthetic_spectrogram_frames(batch=4, seed=3).astype("float64")
        t0 = time.perf_counter()
        for f in sf:
            O.spectrogram_transform(f)
        cpu["spectrogram_prep_samples_per_sec"] = round(4 / (time.perf_counter() - t0), 2)

    if rank == 0:
        # SURVEY 8(a) row C': the deeper EEG-only variant (EEGNetAttentionDeep), eager train steps on the same EEG batch
        deep = brainxai.set_compute_dtype(brainxai.EEGNetAttentionDeep(6, Chans=CHANS, Samples=T), cdt).to(dev).train()
        dopt = brainxai.FlatAdamW(deep.parameters(), lr=1e-3)

        def deep_step():
            dopt.zero_grad()
            dl = crit(deep(eeg), labels)
            dl.backward()
            dopt.step()
        for _ in range(3):
            deep_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            deep_step()
        torch.cuda.synchronize()
        extra["eegnet_attention_deep_train_samples_per_sec"] = round(10 * B / (time.perf_counter() - t0), 1)
        del deep, dopt
    if rank == 0:
        line = {"metric": "samples/sec train (multimodal SpectrogramCNN+EEGNet fusion, B=64/GPU, 4x128x256 spectro + 10000x19 EEG)",
                "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "configs[1]: multimodal train step, B=64 per GPU, spec [64,4,128,256] + EEG [64,1,19,2000] (stacked from [64,10000,19])",
                           "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": "AdamW(1e-3) fused flat arena",
                           "loss": "KLDivLoss(mean)", "dropout": 0.5, "params": sum(p.numel() for p in model.parameters())},
                "hip_graph": graph is not None, "final_loss": round(loss_val, 6), "gradcam": gradcam, "stacker_samples_per_sec": round(stacker_sps, 1), "montage_stacker_samples_per_sec": round(montage_sps, 1), "spectrogram_prep_samples_per_sec": round(specprep_sps, 1),
                "roofline": roofline, "cpu_baseline": cpu}
        line.update(extra)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
