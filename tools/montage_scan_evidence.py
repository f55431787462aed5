#!/usr/bin/env python3
"""Why the native-pipeline montage stacker (bx_eeg_montage_stack) keeps one sequential recurrence per row while the benchmark
stacker (bx_eeg_stack_iir) became a chunked scan in round 2.

The montage chain filters with scipy.signal.lfilter on 10th / 12th order transfer-function coefficients (butter(5 | 6, band-pass
0.5-20 Hz at 200 Hz), reference XAI_Multimodality.py:1245-1267).  In that form the recurrence is ill-conditioned: poles at radius
0.995 / 0.998 and coefficient cancellation make the OUTPUT depend on the exact order of the floating-point operations.  This script
evaluates the very same filters in fp64 as (zero-state response per chunk) + (natural response of the carried state) -- the
algebra a parallel scan uses -- and compares with lfilter's sequential result:

    order-5 band-pass: 1.0e-5 .. 2.7e-5 of the output scale      order-6 band-pass: 3.4e-3 .. 9.8e-3

against a parity tolerance of 2e-5 for that row of SURVEY 8(f).  A scan therefore cannot reproduce the reference's numbers for
this chain in ANY precision the GPU has; the kernel keeps scipy's operation order (and is compiled with -ffp-contract=off for the
same reason).  The benchmark stacker's 4th-order low-pass (poles at 0.795) has no such problem: its scan agrees to 1e-7.
"""
import os
import sys

import numpy as np
from scipy.signal import lfilter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_torch as O  # noqa: E402


def chunked(b, a, x, S):
    n = len(a) - 1
    A = np.zeros((n, n)); A[:, 0] = -a[1:]
    for i in range(n - 1):
        A[i, i + 1] = 1
    y, z = np.zeros_like(x), np.zeros(n)
    for c0 in range(0, len(x), S):
        seg = x[c0:c0 + S]
        y0, zf0 = lfilter(b, a, seg, zi=np.zeros(n))
        zz, nat = z.copy(), np.zeros(len(seg))
        for k in range(len(seg)):
            nat[k] = zz[0]; zz = A @ zz
        y[c0:c0 + S] = y0 + nat
        z = zz + zf0
    return y


if __name__ == "__main__":
    (b5, a5), (b6, a6) = O.montage_coeffs()
    x = np.nan_to_num(O.synthetic_frames(batch=1, seed=7)[0].astype(np.float64)[:, 0])
    for name, (b, a) in (("order-5 band-pass (10th-order tf)", (b5, a5)), ("order-6 band-pass (12th-order tf)", (b6, a6)),
                         ("benchmark low-pass (4th-order tf)", O.butter_lowpass_coeffs())):
        ref = lfilter(b, a, x)
        devs = [np.abs(chunked(b, a, x, S) - ref).max() / np.abs(ref).max() for S in (100, 500, 2500)]
        print(f"{name}: max pole radius {np.abs(np.roots(a)).max():.4f}; chunked (S = 100 / 500 / 2500) vs sequential lfilter: " + " ".join(f"{d:.1e}" for d in devs))
