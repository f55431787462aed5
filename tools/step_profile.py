#!/usr/bin/env python3
"""Training steps only (for rocprofv3): the benchmark model and batch, N hipGraph replays and nothing else, so that a
kernel trace of this process divides cleanly into steps.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d OUT -- python3 $REPO/tools/step_profile.py --steps 50
    python3 $REPO/tools/step_profile.py --summarize OUT --steps 50 > profiles/rNN_step_breakdown.txt
"""
import argparse
import glob
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DT = os.environ.get("BX_PROFILE_DTYPE", "bf16")          # label of the summaries (tools/collect_profiles.sh sets it)


def run(steps, dtype="bf16", ddp=False):
    import torch
    import brainxai
    dev = torch.device("cuda", 0)
    wrap = None
    if ddp:                                 # the data-parallel step on a 1-rank RCCL group (one graph: collectives + AdamW captured)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(42)
    B = 64
    g = torch.Generator().manual_seed(42)
    spec = torch.rand(B, 4, 128, 256, generator=g).to(dev)
    eeg = torch.randn(B, 1, 19, 2000, generator=g).to(dev)
    labels = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), generator=g), 6).float().to(dev)
    model = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32).to(dev).train()
    if ddp:
        wrap = brainxai.DataParallel(model)
    opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
    stepper = brainxai.GraphedTrainStep(model, opt, brainxai.KLDivLoss(), ddp=wrap, adopt_inputs=True, strict=True)   # the resident batch IS the static input
    for _ in range(3):                      # eager, capture, first replay
        stepper((eeg, spec), labels)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss, _ = stepper((eeg, spec), labels)
    e1.record()
    torch.cuda.synchronize()
    print(f"steps {steps}  ms/step {e0.elapsed_time(e1) / steps:.4f}  loss {float(loss):.5f}")
    if ddp:
        brainxai.cleanup()


def run_gradcam(steps):
    """`steps` replayed Grad-CAM sweep batches (eval mode, B=64, all classes, maps upsampled): the other half of the metric."""
    import torch
    import brainxai
    dev = torch.device("cuda", 0)
    B = 64
    g = torch.Generator().manual_seed(42)
    spec = torch.rand(B, 4, 128, 256, generator=g).to(dev)
    eeg = torch.randn(B, 1, 19, 2000, generator=g).to(dev)
    model = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).eval()
    sweep = brainxai.GradCamSweep(model, eeg, spec, class_idx="all")
    for _ in range(3):
        sweep(eeg, spec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = sweep(eeg, spec)
    e1.record()
    torch.cuda.synchronize()
    print(f"batches {steps}  ms/batch {e0.elapsed_time(e1) / steps:.4f}  checksum {float(out.sum()):.6g}")


def summarize(out_dir, steps):
    db = sorted(glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True))[-1]
    con = sqlite3.connect(db)
    rows = list(con.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                            "on d.kernel_id = s.id order by d.start"))
    names = sorted({r[0] for r in rows})
    dem = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
    short = {n: re.sub(r"\(.*", "", d).replace("void ", "") for n, d in zip(names, dem)}
    # the last `steps` AdamW launches delimit the replayed steps
    ends = [i for i, r in enumerate(rows) if short[r[0]].startswith("k_adamw")]
    first = ends[-steps - 1] + 1
    sel = rows[first:ends[-1] + 1]
    agg = {}
    for n, s, e in sel:
        a = agg.setdefault(short[n], [0, 0])
        a[0] += 1
        a[1] += e - s
    busy = sum(v[1] for v in agg.values())
    wall = sel[-1][2] - sel[0][1]
    print(f"# {steps} hipGraph-replayed training steps (B=64, {DT}): wall {wall / steps / 1e3:.1f} us/step, kernel-busy {busy / steps / 1e3:.1f} us/step, "
          f"{sum(v[0] for v in agg.values()) / steps:.0f} launches/step")
    print(f"{'kernel':60s} {'launches/step':>13s} {'avg us':>9s} {'us/step':>9s} {'share':>7s}")
    for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:60]:60s} {cnt / steps:13.2f} {tot / cnt / 1e3:9.2f} {tot / steps / 1e3:9.1f} {100.0 * tot / busy:6.1f}%")


def timeline(out_dir, steps, delim="k_adamw"):
    """One line per launch of a step, in launch order: duration averaged over the replayed steps (every step is the same
    sequence), start offset inside the step, grid and workgroup size -- tells the stages of a shared kernel apart."""
    db = sorted(glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True))[-1]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gcol = "d.grid_size_x, d.grid_size_y, d.workgroup_size_x" if "grid_size_x" in cols else "0, 0, 0"
    rows = list(con.execute(f"select s.kernel_name, d.start, d.end, {gcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                            "on d.kernel_id = s.id order by d.start"))
    names = sorted({r[0] for r in rows})
    dem = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
    short = {n: re.sub(r"\(.*", "", d).replace("void ", "") for n, d in zip(names, dem)}
    ends = [i for i, r in enumerate(rows) if short[r[0]].startswith(delim)]
    per = ends[-1] - ends[-2]
    acc = [[0.0, 0.0, 0.0] for _ in range(per)]
    used = 0
    for k in range(steps):
        hi = ends[-1 - k]
        lo = hi - per + 1
        if lo < 0 or ends[-2 - k] != lo - 1:
            continue
        used += 1
        t0 = rows[lo][1]
        for j in range(per):
            r = rows[lo + j]
            acc[j][0] += r[2] - r[1]
            acc[j][1] += r[1] - t0
            acc[j][2] += (rows[lo + j + 1][1] - r[2]) if j + 1 < per else 0.0
    what = "training step" if delim.startswith("k_adamw") else f"replayed batch (closed by {delim}*)"
    print(f"# launch timeline of one {what}, averaged over {used} hipGraph replays: {per} launches")
    print(f"{'#':>3s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s} {'grid':>12s} {'wg':>5s}  kernel")
    lo = ends[-1] - per + 1
    for j in range(per):
        r = rows[lo + j]
        wg = max(int(r[5]), 1)
        print(f"{j:3d} {acc[j][1] / used / 1e3:9.1f} {acc[j][0] / used / 1e3:8.2f} {acc[j][2] / used / 1e3:7.2f} "
              f"{int(r[3]) // wg:6d}x{int(r[4]):<5d} {wg:5d}  {short[r[0]][:70]}")


def pmc_summarize(fetch_dir, write_dir, steps):
    """HBM traffic per launch from two `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this script
    (counters are KiB; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md, HBM section)."""
    def load(d):
        db = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1]
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(rocpd_pmc_event)")]
        q = ("select s.kernel_name, d.start, e.value from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id = d.event_id "
             "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start")
        rows = list(con.execute(q))
        if not rows:
            raise SystemExit(f"no PMC rows in {db} (columns: {cols})")
        return rows
    out = {}
    for which, d, scale in (("fetch", fetch_dir, 2.0 * 1024), ("write", write_dir, 1024.0)):
        rows = load(d)
        names = sorted({r[0] for r in rows})
        dem = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
        short = {n: re.sub(r"\(.*", "", dm).replace("void ", "") for n, dm in zip(names, dem)}
        ends = [i for i, r in enumerate(rows) if short[r[0]].startswith("k_adamw")]
        sel = rows[ends[-steps - 1] + 1:ends[-1] + 1]
        for n, _, v in sel:
            a = out.setdefault(short[n], {"fetch": [0, 0.0], "write": [0, 0.0]})
            a[which][0] += 1
            a[which][1] += float(v) * scale
    tot_f = sum(v["fetch"][1] for v in out.values()) / steps / 1e6
    tot_w = sum(v["write"][1] for v in out.values()) / steps / 1e6
    # provenance: which kernel sources these counters belong to -- bench.py compares the digest with the library it runs
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("_bx_build", os.path.join(ROOT, "multimodal-brain-pattern-identification_xai_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec); spec.loader.exec_module(bld)
    print(f"# provenance: csrc_digest {bld._digest()} collected {time.strftime('%Y-%m-%d', time.gmtime())} commit {os.environ.get('BX_COMMIT', 'unknown')} "
          f"dtype {os.environ.get('BX_PROFILE_DTYPE', 'bf16')}")
    print(f"# HBM traffic of {steps} eager training steps (B=64, {os.environ.get('BX_PROFILE_DTYPE', 'bf16')}), PMC FETCH_SIZE x2 / WRITE_SIZE in separate passes: "
          f"{tot_f:.0f} MB read + {tot_w:.0f} MB written per step")
    print(f"{'kernel':60s} {'launches/step':>13s} {'fetch MB':>10s} {'write MB':>10s}   (per launch)")
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch"][1] + kv[1]["write"][1])):
        nf, nw = max(v["fetch"][0], 1), max(v["write"][0], 1)
        print(f"{k[:60]:60s} {v['fetch'][0] / steps:13.2f} {v['fetch'][1] / nf / 1e6:10.2f} {v['write'][1] / nw / 1e6:10.2f}")


def sq_summarize(d, steps):
    """Per-kernel SQ counters from one `rocprofv3 --kernel-trace --pmc <names...>` pass of this script (eager steps):
    sums over the launches of the last `steps` steps, plus a few ratios."""
    db = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[-1]
    con = sqlite3.connect(db)
    q = ("select s.kernel_name, d.start, p.name, e.value from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc p on e.pmc_id = p.id order by d.start")
    rows = list(con.execute(q))
    names = sorted({r[0] for r in rows})
    dem = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
    short = {n: re.sub(r"\(.*", "", dm).replace("void ", "") for n, dm in zip(names, dem)}
    starts = sorted({r[1] for r in rows if short[r[0]].startswith("k_adamw")})
    lo, hi = starts[-steps - 1], starts[-1]
    agg, counters = {}, []
    for n, st, c, v in rows:
        if not (lo < st <= hi):
            continue
        if c not in counters:
            counters.append(c)
        agg.setdefault(short[n], {}).setdefault(c, 0.0)
        agg[short[n]][c] += float(v)
    print("# per-kernel SQ counters, sums over %d eager training steps (B=64, %s)" % (steps, DT))
    print("%-44s " % "kernel" + " ".join("%14s" % c[-14:] for c in counters))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        print("%-44s " % k[:44] + " ".join("%14.4g" % v.get(c, 0.0) for c in counters))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--summarize", default=None)
    ap.add_argument("--pmc", nargs=2, default=None, metavar=("FETCH_DIR", "WRITE_DIR"))
    ap.add_argument("--sq", default=None, metavar="DIR")
    ap.add_argument("--timeline", default=None, metavar="DIR")
    ap.add_argument("--delim", default="k_adamw", help="kernel-name prefix of the launch that closes a step (timeline mode)")
    ap.add_argument("--gradcam", action="store_true", help="run the configs[3] Grad-CAM sweep batches instead of training steps")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="activation storage type of the profiled step")
    ap.add_argument("--ddp", action="store_true", help="profile the data-parallel step on a 1-rank RCCL group")
    a = ap.parse_args()
    if a.timeline:
        timeline(a.timeline, a.steps, a.delim)
    elif a.sq:
        sq_summarize(a.sq, a.steps)
    elif a.pmc:
        pmc_summarize(a.pmc[0], a.pmc[1], a.steps)
    elif a.summarize:
        summarize(a.summarize, a.steps)
    elif a.gradcam:
        run_gradcam(a.steps)
    else:
        run(a.steps, a.dtype, a.ddp)
