"""CPU, world_size 2, gloo: the data-parallel wrapper's logic (rank-0 broadcast, one all-reduce of the flat
gradient arena, averaged gradients == single-process gradient of the concatenated batch, `module.` prefixed
state_dict).  The compute here is plain torch on CPU tensors -- only the wrapper / optimizer plumbing is under test."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import brainxai


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(12, 16), nn.Tanh(), nn.Linear(16, 6), nn.LogSoftmax(dim=1))


def _worker(rank, world, port, outdir, flat):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    brainxai.setup(rank, world, backend="gloo")
    try:
        net = _net(100 + rank)                       # ranks start from DIFFERENT weights
        ddp = brainxai.DataParallel(net)
        after_bcast = torch.cat([p.detach().flatten() for p in net.parameters()])
        opt = brainxai.FlatAdamW(net.parameters(), lr=1e-2, allow_host=True) if flat else torch.optim.AdamW(net.parameters(), lr=1e-2)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 12, generator=g)
        y = torch.softmax(torch.randn(8, 6, generator=g), 1)
        xs, ys = x[rank::world], y[rank::world]      # DistributedSampler-style rank-strided shard
        crit = nn.KLDivLoss(reduction="batchmean")
        for _ in range(3):
            opt.zero_grad()
            loss = crit(ddp(xs), ys)
            loss.backward()
            ddp.sync_gradients(opt)
            opt.step(gathered=True) if flat else opt.step()
        keys = list(ddp.state_dict().keys())
        torch.save({"bcast": after_bcast, "final": torch.cat([p.detach().flatten() for p in net.parameters()]), "keys": keys},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        brainxai.cleanup()
        brainxai.ops.clear_grad_views()


@pytest.mark.parametrize("flat", [True, False])
def test_two_rank_gloo_matches_single_process(flat):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, flat), nprocs=world, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world))
    # construction broadcast rank 0's weights
    ref = _net(100)
    want0 = torch.cat([p.detach().flatten() for p in ref.parameters()])
    torch.testing.assert_close(r0["bcast"], want0)
    torch.testing.assert_close(r1["bcast"], want0)
    # replicas stay identical, and equal single-process training on the whole batch (mean of per-shard means ==
    # full-batch mean for equal shard sizes; no BatchNorm in this toy model)
    torch.testing.assert_close(r0["final"], r1["final"])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, generator=g)
    y = torch.softmax(torch.randn(8, 6, generator=g), 1)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    crit = nn.KLDivLoss(reduction="batchmean")
    for _ in range(3):
        opt.zero_grad()
        crit(ref(x), y).backward()
        opt.step()
    want = torch.cat([p.detach().flatten() for p in ref.parameters()])
    torch.testing.assert_close(r0["final"], want, rtol=1e-4, atol=1e-6)
    assert all(k.startswith("module.") for k in r0["keys"])


def test_single_process_wrapper_is_transparent():
    assert not dist.is_initialized()
    net = _net(1)
    ddp = brainxai.DataParallel(net)
    assert ddp.world_size == 1
    x = torch.randn(3, 12)
    torch.testing.assert_close(ddp(x), net(x))
    ddp.sync_gradients(None)     # no-op


class _TwoStage(nn.Module):
    """toy model with the ``cut=`` hook of MultimodalModel.forward: stage A -> (cut) -> stage B"""

    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.a = nn.Sequential(nn.Linear(12, 16), nn.Tanh())
        self.b = nn.Sequential(nn.Linear(16, 6), nn.LogSoftmax(dim=1))

    def forward(self, x, cut=None):
        h = self.a(x)
        if cut is not None:
            h = cut(h)
        return self.b(h)


def _bucket_worker(rank, world, port, outdir):
    from brainxai.train import _Cut
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    brainxai.setup(rank, world, backend="gloo")
    try:
        g = torch.Generator().manual_seed(11)
        x = torch.randn(8, 12, generator=g)
        y = torch.softmax(torch.randn(8, 6, generator=g), 1)
        xs, ys = x[rank::world], y[rank::world]
        crit = nn.KLDivLoss(reduction="batchmean")
        finals = []
        for bucketed in (False, True):
            net = _TwoStage(5)
            ddp = brainxai.DataParallel(net)
            opt = brainxai.FlatAdamW(net.parameters(), lr=1e-2, allow_host=True)
            n_a = sum(p.numel() for p in net.a.parameters())          # arena = [stage A | stage B]
            pidx = len(list(net.a.parameters()))
            for _ in range(3):
                opt.zero_grad()
                if not bucketed:
                    crit(ddp(xs), ys).backward()
                    ddp.sync_gradients(opt)
                else:                                                  # the overlapped step's sequence: late bucket, early bucket
                    cut = _Cut()
                    crit(net(xs, cut=cut), ys).backward()
                    opt.gather_grads(pidx, None)
                    r1 = ddp.reduce_async(opt.flat_g.narrow(0, n_a, opt.n - n_a))
                    cut.finish()
                    opt.gather_grads(0, pidx)
                    r2 = ddp.reduce_async(opt.flat_g.narrow(0, 0, n_a))
                    r1.wait(); r2.wait()
                opt.step(gathered=True)
            finals.append(opt.flat_p.clone())
            opt.close()
        torch.save(finals, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        brainxai.cleanup()
        brainxai.ops.clear_grad_views()


def test_two_bucket_async_reduction_equals_single_collective():
    """the overlapped data-parallel step's plumbing (autograd cut, per-range gather, two asynchronous mean all-reduces of arena
    slices) gives the same weights as one all-reduce after the whole backward, on both ranks"""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_bucket_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world))
    torch.testing.assert_close(r0[0], r0[1], rtol=0, atol=0)
    torch.testing.assert_close(r0[1], r1[1], rtol=0, atol=0)


def test_sharded_sweep_index_arithmetic():
    """configs[3]/[4]: contiguous sample shards per rank, every sample exactly once, ragged tails (see brainxai.sharded_sweep)"""
    for n, world in ((10000, 8), (157, 8), (7, 8), (64, 1), (0, 4)):
        seen = []
        for r in range(world):
            lo, hi = brainxai.shard_bounds(n, r, world)
            assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
