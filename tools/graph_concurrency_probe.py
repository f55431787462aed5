"""Does hipGraph replay run independent branches concurrently on this stack?  Two chains of N small latency-bound kernels
are captured (a) on one stream back to back, (b) forked onto two streams; replay times are compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
N = 200
a = torch.zeros(4096, device=dev); b = torch.zeros(4096, device=dev)


def chain(t):
    for _ in range(N):
        t.add_(1.0)          # tiny kernel: pure launch / latency cost


def timed(g):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    chain(a); chain(b)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    chain(a); chain(b)
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(b)
    chain(a)
    cur.wait_stream(side)
print(f"serial graph  : {timed(g1):.3f} ms for {2 * N} kernels")
print(f"forked graph  : {timed(g2):.3f} ms for {2 * N} kernels")

# (c) two separate graphs replayed on two streams: do different graph launches overlap?
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga):
    chain(a)
with torch.cuda.graph(gb):
    chain(b)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two():
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()


two(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    two()
torch.cuda.synchronize()
print(f"two graphs on two streams: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms for {2 * N} kernels")
# (d) eager on two streams
def eager_two():
    with torch.cuda.stream(s1):
        chain(a)
    with torch.cuda.stream(s2):
        chain(b)
eager_two(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eager_two()
torch.cuda.synchronize()
print(f"eager on two streams     : {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms for {2 * N} kernels (host-bound)")
