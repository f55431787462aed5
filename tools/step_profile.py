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


def run(steps):
    import torch
    import brainxai
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    B = 64
    g = torch.Generator().manual_seed(42)
    spec = torch.rand(B, 4, 128, 256, generator=g).to(dev)
    eeg = torch.randn(B, 1, 19, 2000, generator=g).to(dev)
    labels = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), generator=g), 6).float().to(dev)
    model = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
    opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
    stepper = brainxai.GraphedTrainStep(model, opt, brainxai.KLDivLoss())
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
    print(f"# {steps} hipGraph-replayed training steps (B=64, bf16): wall {wall / steps / 1e3:.1f} us/step, kernel-busy {busy / steps / 1e3:.1f} us/step, "
          f"{sum(v[0] for v in agg.values()) / steps:.0f} launches/step")
    print(f"{'kernel':60s} {'launches/step':>13s} {'avg us':>9s} {'us/step':>9s} {'share':>7s}")
    for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:60]:60s} {cnt / steps:13.2f} {tot / cnt / 1e3:9.2f} {tot / steps / 1e3:9.1f} {100.0 * tot / busy:6.1f}%")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    summarize(a.summarize, a.steps) if a.summarize else run(a.steps)
