"""Would running a layer's data-gradient and weight-gradient kernels side by side pay?  Two hipGraphs (50 dgrads / 50 wgrads
of one stage shape) are replayed back to back on one stream and simultaneously on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brainxai
from brainxai import ops

dev, dt, B = torch.device("cuda:0"), torch.bfloat16, 64
SHAPES = [(128, 256, 16, 16), (64, 128, 32, 32), (32, 64, 64, 64), (16, 32, 128, 128), (8, 16, 256, 256)]


def capture(fn, n, stream):
    with torch.cuda.stream(stream):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(n):
            fn()
    return g


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (H, W, ci, co) in SHAPES:
    x = torch.randn(B, H, W, ci, device=dev).to(dt); dz = torch.randn(B, H, W, co, device=dev).to(dt)
    y = torch.randn(B, H, W, ci, device=dev).to(dt)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05; bias = torch.zeros(co, device=dev)
    pk = ops._pack(w, flip=True, dtype=dt)
    ops.clear_grad_views()
    gd = capture(lambda: ops._conv(dz, pk, None, y, None, False, dt), 50, s1)
    gw = capture(lambda: ops._wgrad(x, dz, w, bias), 50, s2)

    def run(concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            if concurrent:
                with torch.cuda.stream(s1):
                    gd.replay()
                with torch.cuda.stream(s2):
                    gw.replay()
            else:
                with torch.cuda.stream(s1):
                    gd.replay(); gw.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 / 50 * 1e6
    run(False); run(True)
    print(f"B{B} {H}x{W} {ci}->{co}: dgrad+wgrad serial {run(False):6.1f} us   two streams {run(True):6.1f} us")
