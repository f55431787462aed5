#!/usr/bin/env python3
"""Where does the overlapped end-to-end step (brainxai.StagingRing: H2D on a copy stream, stackers on a prep stream, training step
on the main stream) lose time?  Times the legs alone and in combinations on the benchmark batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import brainxai  # noqa: E402

dev = torch.device("cuda:0")
B = 64
g = torch.Generator().manual_seed(42)
raw_h = torch.randn(B, 10000, 19, generator=g) * 100
sraw_h = torch.exp(torch.randn(B, 320, 400, generator=g))
labels = torch.softmax(torch.randn(B, 6, generator=g), 1).to(dev)
torch.manual_seed(42)
model = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
opt = brainxai.FlatAdamW(model.parameters(), lr=1e-3)
eeg = brainxai.stack_eeg(raw_h.to(dev)); spec = brainxai.stack_spectrogram_regions(sraw_h.to(dev))
stepper = brainxai.GraphedTrainStep(model, opt, brainxai.KLDivLoss(), adopt_inputs=True, strict=True)
for _ in range(3):
    stepper([eeg, spec], labels)
sync = torch.cuda.synchronize


def timed(fn, n=40):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3


NSLOT = int(os.environ.get("NSLOT", "4"))


def make_ring(transform):
    ring = brainxai.StagingRing({"eeg": (B, 10000, 19), "spec": (B, 320, 400)}, transform=transform, slots=NSLOT, device=dev,
                                threaded=os.environ.get("THREADED", "1") == "1")
    for _ in range(NSLOT - 1):
        sl = ring.acquire(); sl.host["eeg"].copy_(raw_h); sl.host["spec"].copy_(sraw_h); ring.submit(sl)
    return ring


stack = lambda d: (brainxai.stack_eeg(d["eeg"]), brainxai.stack_spectrogram_regions(d["spec"]))
if "--trace" in sys.argv:                 # a few iterations of the full pipeline only (for rocprofv3)
    ring = make_ring(stack)
    for _ in range(12):
        b = ring.pop(); stepper(list(b.outputs), labels); ring.release(b); ring.submit(ring.acquire())
    sync()
    sys.exit(0)
print(f"step alone (resident inputs)            {timed(lambda: stepper([eeg, spec], labels)):.3f} ms")
print(f"stackers alone (one stream)             {timed(lambda: stack({'eeg': ring0.slots[0].dev['eeg'], 'spec': ring0.slots[0].dev['spec']}) if False else (brainxai.stack_eeg(raw_d), brainxai.stack_spectrogram_regions(sraw_d))) if (globals().__setitem__('raw_d', raw_h.to(dev)) or globals().__setitem__('sraw_d', sraw_h.to(dev)) or True) else 0:.3f} ms")
for name, transform, with_step in (("ring: H2D only", (lambda d: (d["eeg"], d["spec"])), False), ("ring: H2D + stackers", stack, False),
                                   ("ring: H2D + stackers + step", stack, True), ("ring: H2D + step (no stackers)", (lambda d: (eeg, spec)), True)):
    ring = make_ring(transform)

    def it():
        b = ring.pop()
        if with_step:
            stepper(list(b.outputs), labels)
        ring.release(b)
        ring.submit(ring.acquire())
    print(f"{name:40s}{timed(it):.3f} ms")
    ring.close()
    del ring
# host cost of one ring iteration without any GPU work to wait for
ring = make_ring(lambda d: (d["eeg"], d["spec"]))
t0 = time.perf_counter()
for _ in range(40):
    b = ring.pop(); ring.release(b); ring.submit(ring.acquire())
print(f"host time of pop/release/acquire/submit {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms (without waiting for the GPU)")
sync()
