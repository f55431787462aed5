"""GPU: the in-launch statistics finalizes (bxTailDesc.sync: last workgroup to arrive sums the partial rows and finalizes)
against the separate-launch form of the same kernels.  Both sum float partial rows in double in row order; only the grouping
of the double sums differs, so results agree to double rounding before the final float conversion."""
import pytest
import torch

import brainxai
from brainxai import _lib as L
from brainxai import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (cin, cout, batch, H, W): 512 partial rows -> two-level tree; 128 rows -> two groups; 16 rows -> one level
CASES = [(4, 16, 16, 128, 128), (16, 32, 8, 32, 64), (128, 256, 4, 8, 16), (64, 128, 6, 16, 32)]


def _run(in_launch, dtype, case, pool, fuse_pool=False, max_rows=1 << 20, fold=0, route=True):
    cin, cout, batch, h, w = case
    old = ops.TAIL_IN_LAUNCH, ops.FUSE_POOL, ops.TAIL_ROUTE
    ops.TAIL_IN_LAUNCH, ops.FUSE_POOL, ops.TAIL_ROUTE = in_launch, fuse_pool, route
    L.check(L.load().bx_set_tree_max_rows(max_rows), "bx_set_tree_max_rows")      # the default policy (0) never finalizes in-launch
    L.check(L.load().bx_set_tail_fold(fold), "bx_set_tail_fold")                  # these tests pin the folded form off unless asked
    try:
        torch.manual_seed(3)
        blk = brainxai.Block(cin, cout, pool, (2, 2), dropout_p=0.25).to(DEV).train()
        blk.compute_dtype = dtype
        ops.manual_seed(77, DEV)
        outs = []
        for it in range(3):                      # three calls through the same counter words: they must come back to zero
            x = torch.randn(batch, cin, h, w, generator=torch.Generator().manual_seed(5 + it)).to(DEV).requires_grad_(True)
            blk.zero_grad()
            out = blk(x)
            (out.float() * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
            outs.append([out.detach().float().cpu(), x.grad.cpu()] + [p.grad.detach().clone().cpu() for p in blk.parameters()]
                        + [blk.bn.running_mean.clone().cpu(), blk.bn.running_var.clone().cpu(), blk.bn.num_batches_tracked.clone().cpu()])
        torch.cuda.synchronize()
        if in_launch:
            assert blk._sync is not None and int(blk._sync.abs().sum()) == 0, "counter words must be left at zero"
        else:
            assert blk._sync is None
        return outs
    finally:
        ops.TAIL_IN_LAUNCH, ops.FUSE_POOL, ops.TAIL_ROUTE = old
        L.load().bx_set_tree_max_rows(0)
        L.load().bx_set_tail_fold(3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_in_launch_finalize_equals_separate_launches(case, dtype):
    pool = "max" if case[0] % 3 else "avg"
    a = _run(True, dtype, case, pool)
    b = _run(False, dtype, case, pool)
    _compare(a, b, dtype)


@pytest.mark.parametrize("pool", ["max", "avg"])
@pytest.mark.parametrize("case", CASES + [(16, 16, 5, 18, 34), (32, 64, 3, 9, 50)])
def test_conv3_pooled_epilogue_equals_pooling_kernel(case, pool):
    """bf16: conv3 with the 2x2 pool + batch statistics in its epilogue (bx_block_conv3_tail_fwd) against conv3 followed by the
    pooling kernel; pooled values are bit-identical by construction (same stored inputs, same order of additions), the statistics
    differ by the grouping of the double sums only.  The ragged cases have odd H/W remainders and tiles that leave the image."""
    a = _run(True, torch.bfloat16, case, pool, fuse_pool=True)
    b = _run(True, torch.bfloat16, case, pool, fuse_pool=False)
    _compare(a, b, torch.bfloat16)
    c = _run(True, torch.bfloat16, case, pool, fuse_pool=True, max_rows=0)       # pooled epilogue writes rows, separate finalize launch
    _compare(c, b, torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fuse_pool", [False, True])
@pytest.mark.parametrize("case", CASES + [(16, 16, 5, 18, 34), (16, 16, 3, 20, 24)])
def test_folded_finalize_equals_finalize_launches(case, fuse_pool, dtype):
    """round 3: no k_bn_finalize / k_tail_bwd_mid launch -- every workgroup of the apply kernels sums the partial rows itself
    (bx_rows_total).  Same float rows, double sums in another grouping (and, in the backward, fewer rows from the reduction role)."""
    if fuse_pool and dtype != torch.bfloat16:
        pytest.skip("the pooled conv3 epilogue is a bf16 kernel")
    pool = "max" if case[0] % 3 else "avg"
    a = _run(False, dtype, case, pool, fuse_pool=fuse_pool, max_rows=0, fold=3)
    b = _run(False, dtype, case, pool, fuse_pool=fuse_pool, max_rows=0, fold=0)
    _compare(a, b, dtype)
    for mask in (1, 2):                                                           # each direction alone
        _compare(_run(False, dtype, case, pool, fuse_pool=fuse_pool, max_rows=0, fold=mask), b, dtype)


@pytest.mark.parametrize("pool", ["max", "avg"])
@pytest.mark.parametrize("case", CASES + [(16, 16, 5, 18, 34), (32, 64, 3, 9, 50)])
def test_route_nibbles_equal_the_stored_conv3_output(case, pool):
    """round 3: the fused conv3 + pool launch writes WHERE each pooled element's gradient goes (one nibble per pooled element,
    bxTailDesc.route) and does not store conv3's full-resolution output; the backward reads the nibbles.  Same decisions as reading
    the stored output (arg-max with ATen's first-maximum rule and the ReLU test for the max pool, the four ReLU tests for the
    average pool), so every result is bit-identical -- including maps with odd sizes and tiles that leave the image."""
    a = _run(False, torch.bfloat16, case, pool, fuse_pool=True, max_rows=0, fold=3, route=True)
    b = _run(False, torch.bfloat16, case, pool, fuse_pool=True, max_rows=0, fold=3, route=False)
    for ra, rb in zip(a, b):
        for k, (ta, tb) in enumerate(zip(ra, rb)):
            assert torch.equal(ta, tb), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eeg_folded_finalizes_equal_finalize_launches(dtype):
    """round 3: BatchNorm2 (collapsed front end, bf16 storage) and BatchNorm3 of EEGNet are finalized inside the pooling kernels
    (k_eeg_bn_elu_pool sums the partial rows itself); against the k_bn_finalize launches: outputs, input-free parameter gradients
    and running statistics agree to the rounding of the double sums."""
    res = []
    for mask in (3, 0):
        L.check(L.load().bx_set_tail_fold(mask), "bx_set_tail_fold")
        try:
            torch.manual_seed(4)
            net = brainxai.set_compute_dtype(brainxai.EEGNet(6, Chans=19, Samples=2000, dropoutRate=0.25), dtype).to(DEV).train()
            ops.manual_seed(5, DEV)
            x = torch.randn(8, 1, 19, 2000, generator=torch.Generator().manual_seed(6)).to(DEV)
            out = net(x)
            (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
            torch.cuda.synchronize()
            res.append([out.detach().cpu()] + [p.grad.detach().cpu() for p in net.parameters()]
                       + [b.detach().clone().cpu().float() for b in net.buffers()])
        finally:
            L.load().bx_set_tail_fold(3)
    for k, (ta, tb) in enumerate(zip(*res)):
        scale = float(tb.abs().max()) + 1e-30
        assert float((ta - tb).abs().max()) / scale <= 2e-6, k


def _compare(a, b, dtype):
    for it, (ra, rb) in enumerate(zip(a, b)):
        for k, (ta, tb) in enumerate(zip(ra, rb)):
            ta, tb = ta.double(), tb.double()
            scale = float(tb.abs().max()) + 1e-30
            err = float((ta - tb).abs().max()) / scale
            # statistics differ by double-rounding only; in bf16 a one-ulp change of (scale, shift) may move a few stored bf16 values,
            # and every gradient downstream is a sum over those
            tol = 1e-6 if dtype == torch.float32 else 2e-2
            assert err <= tol, (it, k, err)
            if dtype == torch.bfloat16:
                if k == 0:
                    assert float((ta != tb).double().mean()) < 1e-3, (it, k)
                else:
                    assert float((ta - tb).norm() / (tb.norm() + 1e-30)) < 2e-3, (it, k)
