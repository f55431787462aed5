#!/usr/bin/env python3
"""Which input seeds give a flip-free GPU forward?  For each candidate spectrogram seed the multimodal model (train mode,
fp32 storage) runs once on the GPU with ops.keep_block_activations; its post-ReLU activations are compared with the oracle's
fp64 trace (oracle.ref_torch.activation_flips).  Used to choose the inputs of tests/golden/mm_native_small.npz so that the
step-0 gradient comparison is strict (see oracle/make_golden.py: MM_CASES).

    python tools/flip_scan.py 37 3000 3 100 75 4 326 339 335 321 ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import brainxai  # noqa: E402
from brainxai import ops  # noqa: E402
from oracle import ref_torch as O  # noqa: E402
from tests.golden_util import describe_flips, observed_flips  # noqa: E402

chans, samples, cin, h, w, b = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
ref = O.fill_params(O.build_multimodal(chans, samples, cin, dropout=0.0), seed=41).train()
mine = brainxai.build_multimodal(chans, samples, cin, dropout=0.0)
mine.load_state_dict(ref.state_dict())
mine.to(dev).train()
eeg = O.seeded((b, 1, chans, samples), 42, "randn")
for seed in (int(v) for v in sys.argv[7:]):
    spec = O.seeded((b, cin, h, w), seed, "rand")
    keep = ops.keep_block_activations(mine)
    mine.load_state_dict(ref.state_dict())
    with torch.no_grad():
        mine(eeg.to(dev), spec.to(dev))
    torch.cuda.synchronize()
    try:
        flips = observed_flips(O, ref, (eeg, spec), keep, f"seed {seed}")
        print(f"seed {seed}: {len(flips)} flips {describe_flips(flips)}", flush=True)
    except AssertionError as exc:
        print(f"seed {seed}: ERROR {exc}", flush=True)
