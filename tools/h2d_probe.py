import os, sys, time, torch
dev = torch.device("cuda:0")
print("affinity", sorted(os.sched_getaffinity(0))[:4], "...", len(os.sched_getaffinity(0)))
n = 48_640_000 // 4
slots = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(4)]
devb = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4)]
for h in slots: h.normal_()
st = torch.cuda.Stream()
def bw(pairs, reps=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(reps):
            for h, d in pairs: d.copy_(h, non_blocking=True)
        e1.record(st)
    torch.cuda.synchronize()
    return reps * sum(h.numel() * 4 for h, _ in pairs) / (e0.elapsed_time(e1) * 1e-3) / 1e9
for i in range(4):
    print(f"slot {i} -> dev {i}: {bw([(slots[i], devb[i])]):.1f} GB/s")
print(f"all four slots in turn: {bw(list(zip(slots, devb))):.1f} GB/s")
print(f"slot 0 -> four device buffers: {bw([(slots[0], d) for d in devb]):.1f} GB/s")
# separate events per copy (as the ring does)
torch.cuda.synchronize(); t0 = time.perf_counter()
for r in range(5):
    for h, d in zip(slots, devb):
        with torch.cuda.stream(st):
            d.copy_(h, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(st)
torch.cuda.synchronize()
print(f"with an event per copy: {20 * n * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
big = torch.empty(4 * n, dtype=torch.float32).pin_memory(); bigd = torch.empty(4 * n, dtype=torch.float32, device=dev)
print(f"one 194 MB copy: {bw([(big, bigd)], 3):.1f} GB/s")
