#!/usr/bin/env python3
"""Headline benchmark: multimodal (SpectrogramCNN + EEGNet + fusion) TRAINING samples/sec on synthetic
[64,4,128,256] spectrograms + [64,10000,19] raw EEG per GPU (BASELINE.json configs[1]; configs[2] under
--gpus 8), the Grad-CAM sweep of configs[3] (10 000 samples, all classes), the dominant kernel's roofline and the CPU
oracle timed beside it.

    python bench.py                                   # 1 GPU, 100 timed steps, finishes in a couple of minutes
    python bench.py --gpus N --steps K --warmup W     # N > 1: starts N rank processes itself (torch.distributed.run), or
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # ... is started by a launcher, one rank per GPU

A step = zero_grad -> forward -> KLDiv -> backward -> (RCCL all-reduce of the flat gradient arena, in two buckets overlapped
with the early stages' backward) -> fused AdamW over one B=64 batch per rank, inputs resident in HBM, replayed from hipGraphs
by brainxai.GraphedTrainStep -- the same object the product's epoch loops use.  Prints ONE JSON line on rank 0.
"""
import argparse
import glob
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
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense peaks (MI355X_MICROARCH.md); fp32 storage runs on the bf16 pipe, 6 products per FLOP pair
B, CIN, H, W, CHANS, RAW_LEN, T = 64, 4, 128, 256, 19, 10000, 2000
SWEEP_SAMPLES = 10000          # configs[3]
# per Grad-CAM sample (eval, BN folded) forward 7.85 M elements = 15.7 MB bf16 (31.4 MB fp32), 1.90 GFLOP; plus the six
# upsampled fp32 maps it writes (6 x 128 x 256 x 4 B)
GRADCAM_BYTES = {"bf16": 15.7e6 + 6 * H * W * 4, "f32": 31.4e6 + 6 * H * W * 4}
GRADCAM_FLOPS = 1.90e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gradcam", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (stackers, pre-processing, IG, EEGNetAttentionDeep)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying captured hipGraphs")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-storage legs (training step and Grad-CAM sweep of the parity-grade path)")
    return ap.parse_args()


def conv_work(bx, dtype_bytes):
    """Algorithmic FLOPs / bytes of every conv3x3 launch of one training step: forward 15 launches,
    data-gradient 14 (block1.conv1's input needs no gradient), weight-gradient 15.  TRUE channel counts (block1.conv1 has
    4 input planes; the kernels pad them to 8, which is their business).  Returns kind -> [flops, bytes, launches]."""
    stages = [(CIN, 16, 128, 256), (16, 32, 64, 128), (32, 64, 32, 64), (64, 128, 16, 32), (128, 256, 8, 16)]
    out = {"fwd": [0.0, 0.0, 0], "dgrad": [0.0, 0.0, 0], "wgrad": [0.0, 0.0, 0]}
    for si, (cin, c, h, w) in enumerate(stages):
        px = bx * h * w
        for k, ci in enumerate((cin, c, c)):
            fl = 2.0 * 9 * ci * c * px
            by = px * (ci + c) * dtype_bytes
            out["fwd"][0] += fl; out["fwd"][1] += by; out["fwd"][2] += 1
            out["wgrad"][0] += fl; out["wgrad"][1] += by; out["wgrad"][2] += 1
            if not (si == 0 and k == 0):
                # + the ReLU-mask source (conv2/3) or the skip-path addend (conv1) read by the fused epilogue
                out["dgrad"][0] += fl; out["dgrad"][1] += by + px * ci * dtype_bytes; out["dgrad"][2] += 1
    return out


def pmc_conv_traffic(dtype):
    """HBM bytes per conv3x3-family layer launch from the newest committed PMC summary of this dtype (profiles/*_pmc_hbm_traffic*.txt
    written by tools/step_profile.py --pmc: same model, batch and dtype as this benchmark, FETCH_SIZE x2 + WRITE_SIZE collected in
    separate rocprofv3 passes as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process, so the line
    names the file and the provenance the file itself records (collection date, commit, digest of csrc/ at collection time) and says
    whether that digest is the digest of the kernel sources this run was built from.  (bytes, source, matches) or (None, None, None)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bx_build", os.path.join(ROOT, "multimodal-brain-pattern-identification_xai_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec); spec.loader.exec_module(bld)
    now = bld._digest()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic*.txt")))
    fam = ("k_conv_mfma", "k_conv12", "k_conv3x3", "k_conv_split", "k_wgrad_mfma", "k_wgrad_own", "k_wgrad_split", "k_wgrad_direct", "k_wgrad_reduce")
    for path in reversed(files):
        tot, launches, prov = 0.0, 0.0, {}
        for line in open(path):
            if line.startswith("# provenance:"):
                w_ = line.split()
                prov = {w_[i]: w_[i + 1] for i in range(2, len(w_) - 1, 2)}
            if not line.startswith(fam):
                continue
            try:
                wr, fe, n = float(line.split()[-1]), float(line.split()[-2]), float(line.split()[-3])
            except ValueError:
                continue
            tot += n * (fe + wr) * 1e6
            if not line.startswith("k_wgrad_reduce"):
                launches += 2 * n if line.startswith("k_conv12") else n        # a pair launch runs two of the family's 44 layer launches
        if prov.get("dtype", "bf16") != dtype:
            continue
        if launches > 0:
            matches = prov.get("csrc_digest") == now if prov else None
            src = f"profiles/{os.path.basename(path)} (collected {prov.get('collected', 'date not recorded')}, commit {prov.get('commit', 'not recorded')})"
            return round(tot / launches), src, matches
    return None, None, None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes through torch.distributed.run -- BEFORE this
    process makes any GPU call (it never does: a process that has initialised the GPU must not exec or fork GPU workers) -- and
    relay rank 0's JSON line.  Exit code = the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BX_BENCH_FORCE_LAUNCH", None)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)        # stderr of the ranks passes through
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(f"[bench] {args.gpus}-rank run failed (exit code {proc.returncode}, JSON line {'missing' if line is None else 'present'})\n")
        sys.exit(proc.returncode or 1)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    sys.exit(0)


def count_gpus_without_hip():
    """GPUs of this node from the KFD topology in sysfs (nodes with simd_count > 0, minus what ROCR_/HIP_VISIBLE_DEVICES hides):
    the launcher parent must not initialise the GPU before it starts the rank processes, and torch.cuda.device_count() only
    avoids that when its amdsmi path is taken.  None when sysfs has no topology (then the ranks themselves report a shortage)."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def timed(fn, reps, sync):
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return time.perf_counter() - t0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("BX_BENCH_FORCE_LAUNCH") == "1"):
        have = count_gpus_without_hip()                  # sysfs only: this parent never touches HIP
        if have is not None and have < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        launch_ranks(args)
    # stdout carries exactly ONE JSON line: library banners (RCCL prints its version to stdout at init) and any
    # other chatter are routed to stderr for the whole run; the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sync = torch.cuda.synchronize
    force_dist = os.environ.get("BX_BENCH_FORCE_DIST") == "1"      # exercise the data-parallel path with a 1-rank RCCL group
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import brainxai
    from brainxai import ops
    brainxai_stack_regions = brainxai.stack_spectrogram_regions
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    extra = {}

    def all_max(seconds):
        if world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt)
        return seconds

    # ---- synthetic inputs, seed 42 + rank, generated on the host like a DataLoader would
    g = torch.Generator().manual_seed(42 + rank)
    # spectrogram side: parquet-like power values [B, 320 time rows, 4 regions x 100 bins] (positive, heavy-tailed, 0.2 % NaNs)
    # -> the GPU region stacker (window, NaN fill, min-max, anti-aliased resize) -> [B,4,128,256], resident
    sraw = torch.exp(torch.randn(B, 320, 400, generator=g) * 1.5 + torch.linspace(2, -2, 400)[None, None, :])
    sraw[torch.rand(B, 320, 400, generator=g) < 2e-3] = float("nan")
    sraw = sraw.to(dev)
    spec = brainxai_stack_regions(sraw)
    raw = (torch.randn(B, RAW_LEN, CHANS, generator=g) * 100.0)
    flat = raw.view(-1)
    k = flat.numel() // 1000
    idx = torch.randint(0, flat.numel(), (2 * k,), generator=g)
    flat[idx[:k]] = float("nan"); flat[idx[k:]] *= 50.0
    raw = raw.to(dev)
    labels = torch.softmax(torch.randn(B, 6, generator=g), 1).to(dev)
    eeg = brainxai.stack_eeg(raw)                      # [B,1,19,2000], resident
    stacker_sps = montage_sps = specprep_sps = None
    region_sps = 10 * B / timed(lambda: brainxai_stack_regions(sraw), 10, sync)
    if not args.no_extras:
        stacker_sps = 10 * B / timed(lambda: brainxai.stack_eeg(raw), 10, sync)
        # the notebook's native montage chain, raw frames [B,10000,20] -> [B,1,37,3000]
        frames = torch.randn(B, 10000, 20, device=dev) * 50
        brainxai.stack_eeg_montage(frames)
        montage_sps = 3 * B / timed(lambda: brainxai.stack_eeg_montage(frames), 3, sync)
        del frames
        # the notebook's native spectrogram chain, parquet values [B,320,400] -> [B,3,400,300]
        sframes = torch.rand(B, 320, 400, device=dev) * 40
        brainxai.preprocess_spectrograms(sframes)
        specprep_sps = 3 * B / timed(lambda: brainxai.preprocess_spectrograms(sframes), 3, sync)
        del sframes

    torch.manual_seed(42)
    model = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.5, compute_dtype=cdt).to(dev).train()
    ddp = brainxai.DataParallel(model) if (world > 1 or force_dist) else None
    opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
    crit = brainxai.KLDivLoss()
    # the product's own step object: first call eager, second captured (one hipGraph; data-parallel: two, with the first
    # gradient bucket's all-reduce between them), later calls replayed.  The benchmark batch IS the graph's static input.
    # strict: a failed capture raises (a silently eager benchmark would report a ~25 % slower number with rc 0)
    stepper = brainxai.GraphedTrainStep(model, opt, crit, ddp=ddp, adopt_inputs=True, strict=True)
    if args.no_graph:
        stepper.enabled = False
    inputs = [eeg, spec]

    def step():
        return stepper(inputs, labels)

    for _ in range(max(args.warmup, 3)):               # >= 3 so that capture and the first replay are outside the timed region
        loss, _ = step()
    graphed = bool(stepper._graphs)
    if not graphed and not args.no_graph and os.environ.get("BX_GRAPH_LOOPS", "1") != "0":
        sys.exit("bench.py: the training step was not captured into a hipGraph (GraphedTrainStep fell back to eager launches)")
    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = all_max(time.perf_counter() - t0)
    loss_val = float(loss)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- the same step with the stackers in the loop and the inputs starting in HOST memory (SURVEY 8(d) "end-to-end" figure):
    # brainxai.StagingRing -- three pinned host slots, H2D on a copy stream, both GPU stackers on a prep stream, the step on this
    # stream; raw EEG [64,10000,19] + raw spectrogram values [64,320,400] = 81 MB cross PCIe per batch.  The pinned slots are filled
    # once (decoding parquet into them is the DataLoader workers' job and is not what this figure measures); every step re-sends one.
    raw_h, sraw_h = raw.cpu(), sraw.cpu()
    ring = brainxai.StagingRing({"eeg": (B, RAW_LEN, CHANS), "spec": (B, 320, 400)},
                                transform=lambda d: (brainxai.stack_eeg(d["eeg"]), brainxai_stack_regions(d["spec"])), slots=4, device=dev)

    def feed(fill):
        sl = ring.acquire()
        if fill:
            sl.host["eeg"].copy_(raw_h); sl.host["spec"].copy_(sraw_h)
        ring.submit(sl)
    for _ in range(3):
        feed(True)                                      # three in flight, the fourth slot is filled by the first feed(...) below

    def e2e():
        batch = ring.pop()
        stepper(list(batch.outputs), labels)            # device-to-device copy into the graph's static inputs, then the replay
        ring.release(batch)
        feed(not ring.slots[ring._next].h2d)            # a slot's first trip fills it; later trips re-send what it holds
    for _ in range(8):                                  # the first H2D out of a freshly pinned buffer runs at a third of the rate
        e2e()
    e2e_n = min(args.steps, 50)
    e2e_elapsed = all_max(timed(e2e, e2e_n, sync))
    extra["end_to_end_samples_per_sec"] = round(world * B * e2e_n / e2e_elapsed, 1)
    # the PCIe leg alone: one slot's host -> device copies
    sl = ring.slots[0]
    e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync(); e0_.record()
    for _ in range(5):
        for name, host in sl.host.items():
            sl.dev[name].copy_(host, non_blocking=True)
    e1_.record(); sync()
    h2d_bytes = sum(t.numel() * t.element_size() for t in sl.host.values())
    extra["end_to_end_note"] = ("inputs start in pinned host memory: raw EEG [64,10000,19] + raw spectrogram values [64,320,400] (%.1f MB per batch) -> H2D on a "
                                "copy stream (measured alone: %.1f GB/s) -> GPU stackers on a prep stream -> the training step; three of four slots in flight "
                                "(brainxai.StagingRing)" % (h2d_bytes / 1e6, 5 * h2d_bytes / (e0_.elapsed_time(e1_) * 1e-3) / 1e9))
    extra["end_to_end_frac_of_resident"] = round(extra["end_to_end_samples_per_sec"] / value, 4)
    # for comparison: everything on one stream, inputs already on the device (round 2's figure)
    def e2e_one_stream():
        inputs[0].copy_(brainxai.stack_eeg(raw))
        inputs[1].copy_(brainxai_stack_regions(sraw))
        step()
    e2e_one_stream()
    extra["end_to_end_one_stream_device_inputs_samples_per_sec"] = round(world * B * e2e_n / all_max(timed(e2e_one_stream, e2e_n, sync)), 1)
    del ring

    # ---- the parity-grade path beside the headline: fp32 STORAGE (logits / loss / Grad-CAM maps within 1e-3 of the fp32 oracle,
    # gradients within 1e-3 of the decision-matched fp64 twin at this very configuration: tests/test_gpu_bench_config.py), its
    # convolutions on the matrix cores with split-bf16 operands (csrc/conv3x3_split.hip).  Same batch, same step object; N=1 only.
    m32 = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.no_fp32:
        torch.manual_seed(42)
        m32 = brainxai.build_multimodal(CHANS, T, CIN, dropout=0.5, compute_dtype=torch.float32).to(dev).train()
        o32 = brainxai.FlatAdamW(m32.parameters(), lr=1e-3)
        st32 = brainxai.GraphedTrainStep(m32, o32, crit, adopt_inputs=True, strict=True)
        for _ in range(3):
            st32(inputs, labels)
        n32 = min(args.steps, 30)
        t32 = timed(lambda: st32(inputs, labels), n32, sync)
        extra["fp32_samples_per_sec"] = round(B * n32 / t32, 1)
        extra["fp32"] = {"samples_per_sec": round(B * n32 / t32, 1), "ms_per_step": round(t32 / n32 * 1e3, 3), "steps": n32, "hip_graph": bool(st32._graphs),
                         "arithmetic": "fp32 storage; 3x3 convolutions as six bf16 MFMA products per fp32 product (operands split h + m + l), "
                                       "fp32-grade results; everything else fp32 VALU",
                         "step_roofline": {"hbm_frac": round(104.5e6 * B / (t32 / n32) / 1e9 / HBM_PEAK_GBS, 5),
                                           "mfma_frac": round(5.71e9 * B / (t32 / n32) / 1e12 / (MFMA_PEAK_TFLOPS["bf16"] / 6), 5)}}
        st32.release(); o32.close()
        del st32, o32

    # ---- per-kernel HIP-event timing of the conv family: the same step launched eagerly right after the timed
    # region (events cannot bracket kernels inside a replayed graph), same buffers, same data
    prof = []
    was_enabled, stepper.enabled = stepper.enabled, False
    step()
    ops.CONV_PROFILE = []                 # one discarded step in profiling mode
    step()
    ops.CONV_PROFILE = prof
    prof_steps = min(args.steps, 10)
    # Events bracket a launch on the GPU's clock: if the host is the slower side (eager Python between forward launches) the GPU
    # reaches `record(ev0)` early, idles until the kernel arrives, and the idle time lands in the interval (forward convolutions read
    # 0.36-0.45 ms per step from run to run, 0.33 ms in the graph-replay timeline).  ~2.5 ms of device copies queued in front of every
    # measured step keep the GPU behind the host, so the intervals are kernel durations.
    ballast = torch.empty(2, 1 << 30, dtype=torch.uint8, device=dev)
    for _ in range(prof_steps):
        for _ in range(6):
            ballast[1].copy_(ballast[0], non_blocking=True)
        step()
    # What an event pair adds to the kernel it brackets, measured with a real kernel under the same ballast: the interval around ONE
    # small device copy (I1) and around TWO of them back to back (I2); the second copy adds exactly its own duration, so the fixed
    # part of a bracket -- timestamp packets, dispatch after the first, completion before the second -- is 2 * I1 - I2 (medians of
    # 100 brackets each).  It is subtracted from every measured interval below; with it the family's time agrees with the rocprofv3
    # kernel trace of the replayed step (profiles/*_step_timeline.txt), without it every launch reads ~3 us long.
    cal_src, cal_dst = torch.empty(8 << 20, dtype=torch.uint8, device=dev), torch.empty(8 << 20, dtype=torch.uint8, device=dev)
    cal = {1: [], 2: []}
    for rep in range(100):
        if rep % 10 == 0:
            for _ in range(6):
                ballast[1].copy_(ballast[0], non_blocking=True)
        for nk in (1, 2):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            for _ in range(nk):
                cal_dst.copy_(cal_src, non_blocking=True)
            eb.record()
            cal[nk].append((ea, eb))
    del ballast
    sync()
    med = lambda prs: sorted(a_.elapsed_time(b_) for a_, b_ in prs)[len(prs) // 2] * 1e-3          # noqa: E731  (seconds)
    ev_overhead = max(2.0 * med(cal[1]) - med(cal[2]), 0.0)
    del cal_src, cal_dst
    ops.CONV_PROFILE = None
    stepper.enabled = was_enabled

    # ---- dominant kernel family (conv3x3 forward / data gradient / weight gradient): HIP-event durations around each launch of
    # the SAME launches the timed region replays (the fused ones included: conv1 + conv2 pairs of stages 1-3, conv3 with the
    # pooled epilogue -- the library records the event pair around that kernel itself, bx_profile_next_conv3)
    kinds, groups = {}, {}
    dbytes = 2 if args.dtype == "bf16" else 4
    mfma_per_product = 1 if args.dtype == "bf16" else 6      # fp32 storage: six bf16 MFMAs per product (csrc/conv3x3_split.hip)
    for kind, ev0, ev1, meta in prof:
        dt_s = max(ev0.elapsed_time(ev1) * 1e-3 - ev_overhead, 0.0)
        kinds.setdefault(kind, []).append(dt_s)
        form, mb, mh, mw, ci, co = meta
        px = mb * mh * mw
        if form == "pair":                                   # two layers in one launch: both layers' algorithmic bytes and FLOPs
            c1, c2 = co
            cin_true = CIN if ci == 8 and mh == H else ci
            by = px * ((cin_true + c1) + (c1 + c2)) * dbytes
            fl = 2.0 * 9 * px * (cin_true * c1 + c1 * c2)
            name, layers = f"{kind} pair {cin_true}->{c1}->{c2} @{mh}x{mw}", 2
        else:
            cin_true = CIN if (kind != "dgrad" and ci == 8 and mh == H) else ci
            by = px * (cin_true + co) * dbytes + (px * co * dbytes if kind == "dgrad" else 0) + (px // 4 * co * dbytes if form == "conv3+pool" else 0)
            fl = 2.0 * 9 * px * cin_true * co
            name, layers = f"{kind} {'conv3+pool ' if form == 'conv3+pool' else ''}{cin_true}->{co} @{mh}x{mw}", 1
        g_ = groups.setdefault(name, [0.0, 0, by, fl, layers])
        g_[0] += dt_s; g_[1] += 1
    work = conv_work(B, dbytes)
    conv_time = sum(sum(v) for v in kinds.values()) / prof_steps            # seconds per step in conv kernels
    conv_flops = sum(w_[0] for w_ in work.values())
    conv_bytes = sum(w_[1] for w_ in work.values())
    n_launch = sum(w_[2] for w_ in work.values())                           # layer launches of the family (a pair launch runs two)
    n_kernels = sum(len(v) for v in kinds.values()) / max(prof_steps, 1)
    roofline = None
    if conv_time > 0:
        ach_gbs = conv_bytes / conv_time / 1e9
        ach_tf = conv_flops / conv_time / 1e12
        ai = conv_flops / conv_bytes
        peak_tf = MFMA_PEAK_TFLOPS["bf16"] / mfma_per_product               # fp32-storage FLOPs run as 6 bf16 MFMA products each
        ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        traffic, source, traffic_current = pmc_conv_traffic(args.dtype)
        # bf16: the family's arithmetic intensity (~150 FLOP/B) is below the ridge (312 FLOP/B): HBM is the binding roof.
        # fp32 storage: twice the bytes, six MFMA products per FLOP pair -> intensity 77 against a ridge of 52: the matrix cores are
        bound = "hbm" if ai < ridge else "mfma"
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1][0])
        dom_t = dom[0] / dom[1]
        roofline = {"bound": bound, "kernel": "conv3x3 fwd+dgrad+wgrad (%d layer launches in %d kernel launches per step)" % (n_launch, round(n_kernels)),
                    "achieved": round(ach_gbs if bound == "hbm" else ach_tf, 2), "peak": HBM_PEAK_GBS if bound == "hbm" else round(peak_tf, 1),
                    "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                    "frac": round((ach_gbs / HBM_PEAK_GBS) if bound == "hbm" else (ach_tf / peak_tf), 5),
                    "traffic": traffic, "traffic_source": source,
                    "traffic_from_this_build": traffic_current,      # True: the PMC pass was taken on exactly these kernel sources; False: stale
                    "traffic_unit": "HBM bytes per layer launch, family average (PMC FETCH_SIZE x2 + WRITE_SIZE; from the committed summary named in traffic_source, not measured in this run)",
                    "algorithmic_bytes_per_launch": round(conv_bytes / max(n_launch, 1)),
                    "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                    "avg_launch_us": round(conv_time / max(n_launch, 1) * 1e6, 2),
                    "share_of_step": round(conv_time / (elapsed / args.steps), 3),
                    "dominant_kernel": {"name": dom_name, "launches_per_step": round(dom[1] / prof_steps, 1), "avg_us": round(dom_t * 1e6, 2),
                                        "share_of_step": round(dom[0] / prof_steps / (elapsed / args.steps), 4),
                                        "algorithmic_bytes": round(dom[2]), "hbm_frac": round(dom[2] / dom_t / 1e9 / HBM_PEAK_GBS, 4),
                                        "mfma_frac": round(dom[3] / dom_t / 1e12 / peak_tf, 4)},
                    "event_pair_overhead_us": round(ev_overhead * 1e6, 2),
                    "timing": "HIP events around every convolution launch of %d eagerly launched steps right after the timed region -- the "
                              "launches the timed region replays, fused ones included (conv1 + conv2 pairs, conv3 with the pooled epilogue "
                              "bracketed inside the library call); the fixed cost of a bracket (event_pair_overhead_us = 2 x the interval around one "
                              "small copy - the interval around two, medians of 100) is subtracted from every interval" % prof_steps}
        extra["roofline_mfma"] = {"achieved": round(ach_tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                                  "frac": round(ach_tf / peak_tf, 5),
                                  "note": None if mfma_per_product == 1 else "fp32-storage FLOPs; each runs as 6 bf16 MFMA products, peak = 2500 / 6"}
        extra["conv_ms_per_step"] = {k_: round(sum(v) / prof_steps * 1e3, 3) for k_, v in kinds.items()}
        extra["conv_roofline_by_kind"] = {k_: {"hbm_frac": round(work[k_][1] / (sum(v) / prof_steps) / 1e9 / HBM_PEAK_GBS, 4),
                                               "mfma_frac": round(work[k_][0] / (sum(v) / prof_steps) / 1e12 / peak_tf, 4)}
                                          for k_, v in kinds.items()}
        top = sorted(groups.items(), key=lambda kv: -kv[1][0])[:6]
        extra["conv_top_kernels"] = [{"name": n_, "us": round(g_[0] / g_[1] * 1e6, 2), "per_step": round(g_[1] / prof_steps, 1),
                                      "hbm_frac": round(g_[2] / (g_[0] / g_[1]) / 1e9 / HBM_PEAK_GBS, 4),
                                      "mfma_frac": round(g_[3] / (g_[0] / g_[1]) / 1e12 / peak_tf, 4)} for n_, g_ in top]
    # whole-step algorithmic roofline (5.71 GFLOP and 52.2 MB bf16 / 104.5 MB fp32 per sample)
    per_sample_bytes = 52.2e6 if args.dtype == "bf16" else 104.5e6
    extra["step_roofline"] = {"hbm_frac": round(per_sample_bytes * B / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                              "mfma_frac": round(5.71e9 * B / (elapsed / args.steps) / 1e12 / (MFMA_PEAK_TFLOPS["bf16"] / mfma_per_product), 5)}

    # ---- configs[3]: Grad-CAM sweep over 10 000 samples (this rank's contiguous shard of them), eval mode, target block5,
    # all 6 classes, maps upsampled to 128x256: 156 batches of 64 + one of 16 per 10 000 (the ragged tail has its own capture)
    gradcam = None
    if not args.no_gradcam:
        model.eval()
        lo, hi = brainxai.shard_bounds(SWEEP_SAMPLES, rank, world)
        n_mine = hi - lo
        gg = torch.Generator(device=dev).manual_seed(4200 + rank)
        sweep_spec = torch.rand(n_mine, CIN, H, W, generator=gg, device=dev)
        sweep_eeg = torch.randn(n_mine, 1, CHANS, T, generator=gg, device=dev)
        sweep = brainxai.GradCamSweep(model, sweep_eeg[:B], sweep_spec[:B], class_idx="all")
        checksum = torch.zeros((), dtype=torch.float64, device=dev)

        def run_sweep(accumulate=False):
            nb = 0
            for b0 in range(0, n_mine, B):
                maps = sweep(sweep_eeg[b0:b0 + B], sweep_spec[b0:b0 + B])
                if accumulate:
                    checksum.add_(maps.sum(dtype=torch.float64))
                nb += 1
            return nb
        nb = run_sweep(accumulate=True)                     # also captures the tail batch's graph
        sync()
        if world > 1:
            dist.barrier()
        dt = all_max(timed(run_sweep, 1, sync))
        sps = SWEEP_SAMPLES / dt if world > 1 else n_mine / dt
        eager_dt = timed(lambda: brainxai.grad_cam(model, sweep_eeg[:B], sweep_spec[:B], class_idx="all"), 5, sync)
        gradcam = {"maps_per_sec": round(6 * sps, 1), "samples_per_sec": round(sps, 1), "classes": 6, "target": "spectrogram_model.block5",
                   "workload": "configs[3]: %d samples (%d batches of %d%s per rank), eval mode, all classes, maps upsampled to %dx%d" % (
                       SWEEP_SAMPLES, nb, B, " incl. a ragged tail of %d" % (n_mine % B) if n_mine % B else "", H, W),
                   "mode": "hipGraph replay per batch (GradCamSweep); eager grad_cam() calls: %.0f maps/s" % (world * 5 * B * 6 / eager_dt),
                   "roofline": {"bound": "hbm", "achieved": round(sps / world * GRADCAM_BYTES[args.dtype] / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(sps / world * GRADCAM_BYTES[args.dtype] / 1e9 / HBM_PEAK_GBS, 5),
                                "algorithmic_bytes_per_sample": GRADCAM_BYTES[args.dtype],
                                "mfma_frac": round(sps / world * GRADCAM_FLOPS / 1e12 / (MFMA_PEAK_TFLOPS["bf16"] / mfma_per_product), 5),
                                "note": "per GPU; SURVEY 8(d): forward 15.7 MB (bf16) + 1.90 GFLOP per sample, + the 6 upsampled fp32 maps written"},
                   "map_checksum": float(checksum)}
        # the synthetic training state often leaves the class scores' Grad-CAM maps negative almost everywhere, so the sum of the ReLU'd
        # maps can be ~0: also report the pre-ReLU maps of the first batch (same launches, relu off) so that a zero is not mistaken for
        # an empty result
        raw0 = brainxai.grad_cam(model, sweep_eeg[:B], sweep_spec[:B], class_idx="all", relu=False)
        gradcam["first_batch_pre_relu"] = {"abs_sum": float(raw0.abs().sum(dtype=torch.float64)), "min": float(raw0.min()), "max": float(raw0.max())}
        if m32 is not None:                                 # the same sweep through the fp32-storage path (first 2 048 samples)
            m32.eval()
            n32s = min(n_mine, 2048) // B * B
            sweep32 = brainxai.GradCamSweep(m32, sweep_eeg[:B], sweep_spec[:B], class_idx="all")

            def run_sweep32():
                for b0 in range(0, n32s, B):
                    sweep32(sweep_eeg[b0:b0 + B], sweep_spec[b0:b0 + B])
            run_sweep32()
            gradcam["fp32_samples_per_sec"] = round(n32s / timed(run_sweep32, 1, sync), 1)
            gradcam["fp32_maps_per_sec"] = round(6 * gradcam["fp32_samples_per_sec"], 1)
            del sweep32
            m32.train()
        del sweep, sweep_spec, sweep_eeg
        if not args.no_extras:
            # configs[4] run literally: integrated gradients, 50 steps x B=64 (zero baselines, arg-max target), the 64 samples
            # sharded contiguously over the ranks (sharded_sweep: no collective in the data path; at N=1 one rank takes all 64)
            def run_ig():
                return brainxai.sharded_sweep(lambda a, b: brainxai.integrated_gradients(model, (a, b), None, n_steps=50)[1], B, B,
                                              lambda l, h: (eeg[l:h], spec[l:h]), rank=rank, world=world, gather=False)
            run_ig()
            sync()
            if world > 1:
                dist.barrier()
            ig_dt = all_max(timed(run_ig, 1, sync))
            gradcam["ig50_samples_per_sec"] = round(B / ig_dt, 2)
            gradcam["integrated_gradients"] = {"samples_per_sec": round(B / ig_dt, 2), "ms": round(ig_dt * 1e3, 2),
                                               "workload": "configs[4]: 50 steps x B=64 (3200 forward + input-gradient passes), samples sharded "
                                                           "over %d rank(s), interpolants batched %d per pass" % (world, 1024)}
        model.train()

    # ---- measured device-to-device copy rate (quote the box's own streaming rate next to the 8 TB/s spec)
    if rank == 0:
        src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src); sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); sync()
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
        ce, cs, cl = eeg.cpu(), spec.float().cpu(), labels.cpu()
        O.train_step(ref, ropt, ce, cs, cl)
        t0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            O.train_step(ref, ropt, ce, cs, cl)
        cdt_s = time.perf_counter() - t0
        cpu = {"value": round(args.cpu_steps * B / cdt_s, 2), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{args.cpu_steps} training steps of the same B={B} batch after 1 warm-up step (oracle/ref_torch.py, fp32, "
                         f"{torch.get_num_threads()} torch threads; host shows {os.cpu_count()} CPUs, share is 16)"}
        if gradcam is not None:
            # the oracle takes the product's (trained) weights, so that its maps are also the yardstick for the maps the
            # benchmarked sweep produces: error of the bf16 sweep (and of the fp32-storage one) on the scale of the pre-ReLU maps
            ref.load_state_dict(model.state_dict())
            ref.eval()
            t0 = time.perf_counter()
            cams = [O.grad_cam(ref, ce[16 * i:16 * i + 16], cs[16 * i:16 * i + 16], class_idx="all") for i in range(3)]
            cpu["gradcam_maps_per_sec"] = round(3 * 16 * 6 / (time.perf_counter() - t0), 2)
            cpu["gradcam_sample"] = "3 batches of 16 samples, forward hook + one autograd pass per class"
            raw0 = O.grad_cam(ref, ce[:16], cs[:16], class_idx="all", relu=False)
            scale0 = float(raw0.abs().max())
            model.eval()
            sw16 = brainxai.GradCamSweep(model, eeg[:16], spec[:16], class_idx="all")
            got = sw16(eeg[:16], spec[:16]).float().cpu()
            gradcam["max_rel_err_vs_fp32_oracle"] = float((got.double() - cams[0].double()).abs().max()) / scale0
            gradcam["max_rel_err_note"] = ("maps of the benchmarked %s sweep (GradCamSweep, 16 samples x 6 classes, trained weights) against the CPU fp32 "
                                           "oracle's, relative to the largest pre-ReLU map value" % args.dtype)
            del sw16
            model.train()
            if m32 is not None:
                m32.load_state_dict(model.state_dict())
                m32.eval()
                sw16 = brainxai.GradCamSweep(m32, eeg[:16], spec[:16], class_idx="all")
                got = sw16(eeg[:16], spec[:16]).float().cpu()
                gradcam["fp32_max_rel_err_vs_fp32_oracle"] = float((got.double() - cams[0].double()).abs().max()) / scale0
                del sw16
        t0 = time.perf_counter()
        O.stack_eeg_batch(raw[:16].cpu().numpy())
        cpu["stacker_samples_per_sec"] = round(16 / (time.perf_counter() - t0), 2)                 # one host core, numpy/scipy (dataset.py:73-104)
        sr_host = sraw[:8].cpu().numpy()
        t0 = time.perf_counter()
        for f in sr_host:
            O.spectrogram_regions_transform(f)
        cpu["spectrogram_region_stacker_samples_per_sec"] = round(8 / (time.perf_counter() - t0), 2)
        if not args.no_extras:
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

    if rank == 0 and not args.no_extras:
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
        extra["eegnet_attention_deep_train_samples_per_sec"] = round(10 * B / timed(deep_step, 10, sync), 1)
        dopt.close()
        del deep, dopt
    if rank == 0:
        line = {"metric": "samples/sec train (multimodal SpectrogramCNN+EEGNet fusion, B=64/GPU, 4x128x256 spectro + 10000x19 EEG)",
                "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "configs[1]: multimodal train step, B=64 per GPU, spec [64,4,128,256] + EEG [64,1,19,2000] (stacked from [64,10000,19])",
                           "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": "AdamW(1e-3) fused flat arena",
                           "loss": "KLDivLoss(mean)", "dropout": 0.5, "params": sum(p.numel() for p in model.parameters()),
                           "gradient_exchange": None if ddp is None else ("two arena buckets, all-reduce(AVG) overlapped with the early stages' backward"
                                                                           if stepper.plan is not None else "one all-reduce(AVG) of the flat arena")},
                "hip_graph": graphed, "final_loss": round(loss_val, 6), "gradcam": gradcam,
                "stacker_samples_per_sec": None if stacker_sps is None else round(stacker_sps, 1),
                "spectrogram_region_stacker_samples_per_sec": round(region_sps, 1),
                "montage_stacker_samples_per_sec": None if montage_sps is None else round(montage_sps, 1),
                "spectrogram_prep_samples_per_sec": None if specprep_sps is None else round(specprep_sps, 1),
                "roofline": roofline, "cpu_baseline": cpu}
        line.update(extra)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        # every rank is done and the result line is out.  Teardown through the product's own cleanup(): it releases the graphed
        # step's captured graphs (they contain the RCCL collectives) and any reduction handle before the communicator goes
        dist.barrier()
        brainxai.cleanup()


if __name__ == "__main__":
    main()
