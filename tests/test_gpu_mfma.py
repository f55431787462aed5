"""GPU: the MFMA implicit-GEMM kernels (bf16 operands, fp32 accumulate) against (a) the direct VALU kernel on
the SAME bf16 inputs and bf16-rounded weights -- agreement to fp32-summation noise before the final bf16
rounding -- and (b) a plain PyTorch fp32 reference of the op."""
import pytest
import torch
import torch.nn.functional as F

import brainxai
from brainxai import _lib as L
from brainxai import ops
from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _bf(t):
    return t.to(torch.bfloat16).float()


SHAPES = [(8, 16, 9, 13), (16, 16, 8, 32), (16, 32, 17, 40), (32, 64, 8, 16), (64, 64, 5, 6), (64, 128, 16, 32),
          (128, 256, 8, 16), (256, 256, 4, 8), (128, 128, 25, 18)]


@pytest.mark.parametrize("cin,cout,h,w", SHAPES)
def test_conv_mfma_forward(cin, cout, h, w):
    torch.manual_seed(cin * 131 + cout)
    x = _bf(torch.randn(3, cin, h, w))
    wt = _bf(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5))
    b = torch.randn(cout)
    want = F.relu(F.conv2d(x, wt, b, padding=1))
    xn = ops.to_nhwc(x.to(DEV), torch.bfloat16)
    packed = ops._pack(wt.to(DEV), False)
    assert packed[1] is not None, "MFMA operand must be produced for this shape"
    lib = L.load()
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("mfma", L.BX_ALGO_MFMA)):
        y = torch.empty(3, h, w, cout, dtype=torch.bfloat16, device=DEV)
        L.check(lib.bx_conv3x3(xn.data_ptr(), packed[0].data_ptr(), packed[1].data_ptr(), b.to(DEV).data_ptr(), None, None, y.data_ptr(),
                               3, h, w, cin, cout, L.BX_BF16, L.BX_EPI_RELU, algo, torch.cuda.current_stream().cuda_stream), name)
        outs[name] = ops.to_nchw_f32(y, cout).cpu()
    torch.cuda.synchronize()
    assert rel_err(outs["mfma"], want) < 6e-3            # bf16 output rounding: 2^-8 relative per element
    assert rel_err(outs["mfma"], outs["direct"]) < 6e-3  # both round the same fp32 sums (up to summation order)
    frac_equal = float((outs["mfma"] == outs["direct"]).float().mean())
    assert frac_equal > 0.98, frac_equal


@pytest.mark.parametrize("cin,cout,batch", [(128, 256, 4), (256, 256, 6), (256, 256, 5)])
def test_conv_mfma_two_image_tiles(cin, cout, batch):
    """stage-5 shapes (the 8 x 16 tile is the whole image, 256 output channels): even batches run two images per workgroup,
    odd batches the one-image kernel; both must equal the direct kernel and the fp32 reference, image by image"""
    torch.manual_seed(cin + cout + batch)
    h, w = 8, 16
    x = _bf(torch.randn(batch, cin, h, w))
    wt = _bf(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5))
    b = torch.randn(cout)
    ymask, add = _bf(torch.randn(batch, cout, h, w)), _bf(torch.randn(batch, cout, h, w))
    want = F.conv2d(x, wt, b, padding=1) * (ymask > 0) + add
    xn, mn, an = (ops.to_nhwc(t.to(DEV), torch.bfloat16) for t in (x, ymask, add))
    packed = ops._pack(wt.to(DEV), False)
    lib = L.load()
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("mfma", L.BX_ALGO_MFMA)):
        y = torch.empty(batch, h, w, cout, dtype=torch.bfloat16, device=DEV)
        L.check(lib.bx_conv3x3(xn.data_ptr(), packed[0].data_ptr(), packed[1].data_ptr(), b.to(DEV).data_ptr(), mn.data_ptr(), an.data_ptr(),
                               y.data_ptr(), batch, h, w, cin, cout, L.BX_BF16, 0, algo, torch.cuda.current_stream().cuda_stream), name)
        outs[name] = ops.to_nchw_f32(y, cout).cpu()
    torch.cuda.synchronize()
    for i in range(batch):
        assert rel_err(outs["mfma"][i], want[i]) < 8e-3, i
        assert rel_err(outs["mfma"][i], outs["direct"][i]) < 8e-3, i


@pytest.mark.parametrize("cin,cout,h,w", [(16, 16, 12, 20), (32, 16, 9, 33), (64, 32, 8, 16), (128, 64, 6, 7), (256, 128, 4, 8), (256, 256, 8, 16),
                                              (256, 256, 11, 37)])
def test_conv_mfma_data_gradient_epilogue(cin, cout, h, w):
    """flip/transpose pack + ReLU-mask + addend epilogue: dX = conv(dZ, W^T flipped) * (Y > 0) + A"""
    torch.manual_seed(7 + cin)
    co_layer, ci_layer = cin, cout          # the layer maps ci_layer -> co_layer; its data-gradient maps back
    wt = _bf(torch.randn(co_layer, ci_layer, 3, 3) / (3 * ci_layer ** 0.5))
    dz = _bf(torch.randn(2, co_layer, h, w))
    ymask = _bf(torch.randn(2, ci_layer, h, w))
    add = _bf(torch.randn(2, ci_layer, h, w))
    want = F.conv_transpose2d(dz, wt, padding=1) * (ymask > 0) + add
    packed = ops._pack(wt.to(DEV), True)
    assert packed[1] is not None
    dzn, mn, an = (ops.to_nhwc(t.to(DEV), torch.bfloat16) for t in (dz, ymask, add))
    lib = L.load()
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("mfma", L.BX_ALGO_MFMA)):
        y = torch.empty(2, h, w, ci_layer, dtype=torch.bfloat16, device=DEV)
        L.check(lib.bx_conv3x3(dzn.data_ptr(), packed[0].data_ptr(), packed[1].data_ptr(), None, mn.data_ptr(), an.data_ptr(), y.data_ptr(),
                               2, h, w, co_layer, ci_layer, L.BX_BF16, 0, algo, torch.cuda.current_stream().cuda_stream), name)
        outs[name] = ops.to_nchw_f32(y, ci_layer).cpu()
    torch.cuda.synchronize()
    assert rel_err(outs["mfma"], want) < 8e-3
    assert rel_err(outs["mfma"], outs["direct"]) < 8e-3


@pytest.mark.parametrize("co_layer,batch,h,w", [(16, 2, 12, 20), (32, 3, 9, 33), (16, 64, 128, 256)])
def test_conv_mfma_data_gradient_to_8_padded_channels(co_layer, batch, h, w):
    """the data gradient of a stage-1 conv1 (4 input planes padded to 8; only saliency / integrated gradients ask for it): the MFMA
    operand has 16 rows, 8 of them zero, and the kernels store the 8 real channels -- one-shot kernel at the small sizes, persistent
    kernel at the benchmark's batch; must equal the direct kernel and the fp32 reference, and the 4 padding channels must be 0 + addend"""
    torch.manual_seed(11 + co_layer + h)
    ci_layer = 4
    wt = _bf(torch.randn(co_layer, ci_layer, 3, 3) / (3 * ci_layer ** 0.5))
    dz = _bf(torch.randn(batch, co_layer, h, w))
    add = _bf(torch.randn(batch, ci_layer, h, w))
    want = F.conv_transpose2d(dz, wt, padding=1) + add
    packed = ops._pack(wt.to(DEV), True)
    assert packed[1] is not None and packed[3] == 8, "MFMA operand must exist for an 8-channel data gradient"
    dzn, an = (ops.to_nhwc(t.to(DEV), torch.bfloat16) for t in (dz, add))
    assert an.shape[3] == 8
    lib = L.load()
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("mfma", L.BX_ALGO_MFMA)):
        y = torch.full((batch, h, w, 8), 7.0, dtype=torch.bfloat16, device=DEV)
        L.check(lib.bx_conv3x3(dzn.data_ptr(), packed[0].data_ptr(), packed[1].data_ptr(), None, None, an.data_ptr(), y.data_ptr(),
                               batch, h, w, co_layer, 8, L.BX_BF16, 0, algo, torch.cuda.current_stream().cuda_stream), name)
        torch.cuda.synchronize()
        assert float(y[..., 4:].float().abs().max()) == 0.0, name       # padding channels: zero weights + zero addend
        outs[name] = ops.to_nchw_f32(y, ci_layer).cpu()
    assert rel_err(outs["mfma"], want) < 8e-3
    assert rel_err(outs["mfma"], outs["direct"]) < 8e-3


@pytest.mark.parametrize("cin,c1,batch,h,w", [(4, 16, 3, 20, 45), (4, 16, 2, 8, 32), (4, 16, 1, 9, 33), (4, 16, 64, 128, 256),
                                                (16, 32, 3, 20, 45), (16, 32, 2, 8, 32), (16, 32, 1, 9, 33), (16, 32, 64, 64, 128),
                                                (32, 64, 3, 20, 45), (32, 64, 2, 8, 32), (32, 64, 1, 9, 33), (32, 64, 64, 32, 64)])
def test_conv_pair_is_bit_identical_to_two_launches(cin, c1, batch, h, w):
    """stage 1's conv1 + conv2 in one launch (conv1's tile recomputed on the halo and kept in LDS): the same MFMA sequence on the same
    operands as the two separate launches, so y1 and y2 must be EQUAL bit for bit -- with the conv1 output written out (training) and
    without (evaluation-mode passes); image edges, ragged tiles, and the benchmark's batch (several tiles per workgroup)"""
    torch.manual_seed(5 + h)
    x = _bf(torch.rand(batch, cin, h, w))
    w1 = _bf(torch.randn(c1, cin, 3, 3) / (3 * cin ** 0.5))
    w2 = _bf(torch.randn(c1, c1, 3, 3) / (3 * c1 ** 0.5))
    b1, b2 = torch.randn(c1) * 0.1, torch.randn(c1) * 0.1
    cp = ops.pad8(cin)
    xn = ops.to_nhwc(x.to(DEV), torch.bfloat16)
    p1, p2 = ops._pack(w1.to(DEV), False, torch.bfloat16), ops._pack(w2.to(DEV), False, torch.bfloat16)
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    b1d, b2d = b1.to(DEV), b2.to(DEV)
    y1 = torch.empty(batch, h, w, c1, dtype=torch.bfloat16, device=DEV)
    y2 = torch.empty_like(y1)
    L.check(lib.bx_conv3x3(xn.data_ptr(), None, p1[1].data_ptr(), b1d.data_ptr(), None, None, y1.data_ptr(), batch, h, w, cp, c1, L.BX_BF16,
                           L.BX_EPI_RELU, L.BX_ALGO_MFMA, st), "conv1")
    L.check(lib.bx_conv3x3(y1.data_ptr(), None, p2[1].data_ptr(), b2d.data_ptr(), None, None, y2.data_ptr(), batch, h, w, c1, c1, L.BX_BF16,
                           L.BX_EPI_RELU, L.BX_ALGO_MFMA, st), "conv2")
    z1 = torch.full_like(y1, 3.0)
    z2, z2b = torch.full_like(y1, 3.0), torch.full_like(y1, 3.0)
    L.check(lib.bx_conv3x3_pair(xn.data_ptr(), p1[1].data_ptr(), b1d.data_ptr(), p2[1].data_ptr(), b2d.data_ptr(), z1.data_ptr(), z2.data_ptr(),
                                None, None, batch, h, w, cp, c1, c1, L.BX_BF16, st), "pair")
    L.check(lib.bx_conv3x3_pair(xn.data_ptr(), p1[1].data_ptr(), b1d.data_ptr(), p2[1].data_ptr(), b2d.data_ptr(), None, z2b.data_ptr(),
                                None, None, batch, h, w, cp, c1, c1, L.BX_BF16, st), "pair (no y1)")
    torch.cuda.synchronize()
    assert torch.equal(z1.view(torch.int16), y1.view(torch.int16))
    assert torch.equal(z2.view(torch.int16), y2.view(torch.int16))
    assert torch.equal(z2b.view(torch.int16), y2.view(torch.int16))
    if batch <= 3:                                         # and against the fp32 reference of the two layers
        want = F.relu(F.conv2d(_bf(F.relu(F.conv2d(x, w1, b1, padding=1))), w2, b2, padding=1))
        assert rel_err(ops.to_nchw_f32(z2, c1).cpu(), want) < 8e-3


@pytest.mark.parametrize("cin,c1,batch,h,w", [(4, 16, 3, 20, 45), (4, 16, 64, 128, 256), (16, 32, 3, 20, 45), (16, 32, 64, 64, 128)])
def test_relu_bit_masks_give_the_same_data_gradient(cin, c1, batch, h, w):
    """bx_conv3x3_pair also writes the ReLU decisions of y1 / y2 as bits (one byte per pixel and 4 channels); a data gradient that
    reads the bits (BX_EPI_MASK_BITS) must equal, bit for bit, the one that reads the activations -- one-shot kernels at the small
    sizes, the persistent kernel (16 channels) at the benchmark's batch; and the bits themselves must be (y > 0)"""
    torch.manual_seed(9 + h + c1)
    x = _bf(torch.rand(batch, cin, h, w) - 0.3)
    w1 = _bf(torch.randn(c1, cin, 3, 3) / (3 * cin ** 0.5))
    w2 = _bf(torch.randn(c1, c1, 3, 3) / (3 * c1 ** 0.5))
    b1, b2 = (torch.randn(c1) * 0.1).to(DEV), (torch.randn(c1) * 0.1).to(DEV)
    cp = ops.pad8(cin)
    xn = ops.to_nhwc(x.to(DEV), torch.bfloat16)
    p1, p2 = ops._pack(w1.to(DEV), False, torch.bfloat16), ops._pack(w2.to(DEV), False, torch.bfloat16)
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    y1 = torch.empty(batch, h, w, c1, dtype=torch.bfloat16, device=DEV)
    y2 = torch.empty_like(y1)
    m1 = torch.full((batch, h, w, c1 // 4), 255, dtype=torch.uint8, device=DEV)
    m2 = torch.full_like(m1, 255)
    L.check(lib.bx_conv3x3_pair(xn.data_ptr(), p1[1].data_ptr(), b1.data_ptr(), p2[1].data_ptr(), b2.data_ptr(), y1.data_ptr(), y2.data_ptr(),
                                m1.data_ptr(), m2.data_ptr(), batch, h, w, cp, c1, c1, L.BX_BF16, st), "pair")
    torch.cuda.synchronize()
    for y, m in ((y1, m1), (y2, m2)):
        pos = (y.float() > 0).reshape(batch, h, w, c1 // 4, 4).to(torch.uint8)
        want = pos[..., 0] | (pos[..., 1] << 1) | (pos[..., 2] << 2) | (pos[..., 3] << 3)
        assert torch.equal(m, want)
        assert 0.05 < float(pos.float().mean()) < 0.95          # both decisions occur
    # data gradient of conv2 (c1 -> c1): dZ1 = conv(dZ2, W2^T flipped) * (y1 > 0)
    dz = ops.to_nhwc(_bf(torch.randn(batch, c1, h, w)).to(DEV), torch.bfloat16)
    pd = ops._pack(w2.to(DEV), True, torch.bfloat16)
    outs = []
    for mask, flags in ((y1, 0), (m1, L.BX_EPI_MASK_BITS)):
        o = torch.full((batch, h, w, c1), 5.0, dtype=torch.bfloat16, device=DEV)
        L.check(lib.bx_conv3x3(dz.data_ptr(), None, pd[1].data_ptr(), None, mask.data_ptr(), None, o.data_ptr(), batch, h, w, c1, c1, L.BX_BF16,
                               flags, L.BX_ALGO_MFMA, st), "dgrad")
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_block_bf16_mfma_matches_direct_path():
    """whole Block forward+backward in bf16: MFMA kernels vs direct kernels (same storage rounding points)"""
    torch.manual_seed(11)
    res = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("auto", L.BX_ALGO_AUTO)):
        ops.CONV_ALGO = ops.WGRAD_ALGO = algo
        try:
            torch.manual_seed(11)
            blk = brainxai.Block(32, 64, "avg", (2, 2), dropout_p=0.0).to(DEV).train()
            blk.compute_dtype = torch.bfloat16
            x = torch.randn(4, 32, 16, 32, generator=torch.Generator().manual_seed(5)).to(DEV).requires_grad_(True)
            out = blk(x)
            (out.float() * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
            res[name] = (out.detach().float().cpu(), x.grad.cpu(), {n: p.grad.cpu() for n, p in blk.named_parameters()})
        finally:
            ops.CONV_ALGO = ops.WGRAD_ALGO = L.BX_ALGO_AUTO
    torch.cuda.synchronize()
    assert rel_err(res["auto"][0], res["direct"][0]) < 2e-2
    # the two paths round different fp32 sums to bf16, so a few ReLU masks flip: compare dX in relative L2
    dxa, dxd = res["auto"][1].double(), res["direct"][1].double()
    assert float((dxa - dxd).norm() / dxd.norm()) < 0.12
    for n in res["auto"][2]:
        a, d = res["auto"][2][n], res["direct"][2][n]
        cos = float(F.cosine_similarity(a.flatten().double(), d.flatten().double(), dim=0))
        assert cos > 0.98, (n, cos)


WG_SHAPES = [(4, 8, 16, 9, 13), (16, 16, 16, 8, 32), (16, 16, 32, 17, 40), (32, 32, 32, 8, 16), (32, 32, 64, 16, 32),
             (64, 64, 64, 5, 6), (64, 64, 128, 16, 32), (128, 128, 256, 8, 16), (256, 256, 256, 4, 8), (128, 128, 128, 25, 18)]


@pytest.mark.parametrize("cin,cip,cout,h,w", WG_SHAPES)
def test_wgrad_mfma(cin, cip, cout, h, w):
    torch.manual_seed(cin + 3 * cout)
    B = 3
    x = _bf(torch.randn(B, cin, h, w))
    dz = _bf(torch.randn(B, cout, h, w))
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    bias = torch.zeros(cout, requires_grad=True)
    (F.conv2d(x, wt, bias, padding=1) * dz).sum().backward()
    xn = ops.to_nhwc(x.to(DEV), torch.bfloat16)
    assert xn.shape[3] == cip
    dzn = ops.to_nhwc(dz.to(DEV), torch.bfloat16)
    lib = L.load()
    got = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("mfma", L.BX_ALGO_MFMA)):
        need = lib.bx_conv3x3_wgrad_workspace(B, h, w, cip, cout, L.BX_BF16, algo)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_BF16, algo,
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), name)
        got[name] = (dw.cpu(), db.cpu())
    torch.cuda.synchronize()
    for name in got:
        assert rel_err(got[name][0], wt.grad) < 1e-4, name          # bf16 inputs are exact in fp32; only summation order differs
        assert rel_err(got[name][1], bias.grad) < 1e-4, name


def test_wgrad_chained_reduce_matches_immediate():
    """bx_conv3x3_wgrad_chained: a layer's partial sum rides in the next layer's launch; four layers of different tilings in a
    chain + bx_conv3x3_wgrad_finish agree with four bx_conv3x3_wgrad calls (same fixed-order sums up to the slice count,
    hence 1e-6) and leave nothing pending; re-using the pending workspace is refused."""
    import ctypes
    lib = L.load()
    torch.manual_seed(4)
    layers = [(16, 32, 32, 64), (64, 64, 16, 32), (8, 16, 64, 128), (256, 256, 8, 16)]
    B = 4
    data, want = [], []
    for cin, cout, h, w in layers:
        cip = ops.pad8(cin)
        xn = ops.to_nhwc(torch.randn(B, cin, h, w, device=DEV), torch.bfloat16)
        dzn = ops.to_nhwc(torch.randn(B, cout, h, w, device=DEV) * 0.1, torch.bfloat16)
        dw, db = torch.empty(cout, cin, 3, 3, device=DEV), torch.empty(cout, device=DEV)
        need = lib.bx_conv3x3_wgrad_workspace(B, h, w, cip, cout, L.BX_BF16, L.BX_ALGO_MFMA)
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_BF16, L.BX_ALGO_MFMA,
                                     ws.data_ptr(), ws.numel(), 0), "wgrad")
        data.append((xn, dzn, cin, cip, cout, h, w, need))
        want.append((dw, db))
    pend = L.WgradPending()
    got, bufs = [], []
    for xn, dzn, cin, cip, cout, h, w, need in data:
        dw, db = torch.full((cout, cin, 3, 3), float("nan"), device=DEV), torch.full((cout,), float("nan"), device=DEV)
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        bufs.append(ws)
        L.check(lib.bx_conv3x3_wgrad_chained(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_BF16,
                                             L.BX_ALGO_MFMA, ws.data_ptr(), ws.numel(), ctypes.byref(pend), 0), "wgrad chained")
        assert pend.valid == 1
        got.append((dw, db))
    torch.cuda.synchronize()
    assert torch.isnan(got[-1][0]).all()                      # the last layer is still pending ...
    assert not torch.isnan(got[-2][0]).any()                  # ... the one before it was summed inside the last launch
    L.check(lib.bx_conv3x3_wgrad_finish(ctypes.byref(pend), 0), "finish")
    assert pend.valid == 0
    torch.cuda.synchronize()
    for (dw, db), (rw, rb) in zip(got, want):
        assert rel_err(dw.cpu(), rw.cpu()) < 1e-6 and rel_err(db.cpu(), rb.cpu()) < 1e-6
    xn, dzn, cin, cip, cout, h, w, need = data[0]
    dw, db = torch.empty(cout, cin, 3, 3, device=DEV), torch.empty(cout, device=DEV)
    L.check(lib.bx_conv3x3_wgrad_chained(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_BF16,
                                         L.BX_ALGO_MFMA, bufs[0].data_ptr(), bufs[0].numel(), ctypes.byref(pend), 0), "wgrad chained")
    rc = lib.bx_conv3x3_wgrad_chained(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_BF16,
                                      L.BX_ALGO_MFMA, bufs[0].data_ptr(), bufs[0].numel(), ctypes.byref(pend), 0)
    assert rc != 0
    L.check(lib.bx_conv3x3_wgrad_finish(ctypes.byref(pend), 0), "finish")
    torch.cuda.synchronize()


# the benchmark's stage shapes at its full batch: (cin, cout, H, W)
BENCH_LAYERS = [(16, 16, 128, 256), (16, 32, 64, 128), (32, 32, 64, 128), (32, 64, 32, 64), (64, 64, 32, 64), (64, 128, 16, 32),
                (128, 128, 16, 32), (128, 256, 8, 16), (256, 256, 8, 16)]


@pytest.mark.parametrize("cin,cout,h,w", BENCH_LAYERS)
def test_conv_family_adjoint_identity_at_bench_size(cin, cout, h, w):
    """Size-independent property at BASELINE's full batch (B=64), every stage shape, the kernels the benchmark actually launches:
    forward, data-gradient and weight-gradient kernels are three views of ONE trilinear form,
        <conv(x, w), dz>  ==  <x, conv^T(dz, w)>  ==  <w, wgrad(x, dz)>,
    so an indexing slip in any tile variant (two-image tiles, channel-split waves, persistent tiles, split reductions) breaks an
    equality that no oracle run is needed for.  All operands are positive, so the form is of the order of its norm bound and a
    misplaced or missing tile shifts it by its share of the elements; bf16 outputs carry 2^-9 relative rounding per element,
    unbiased, which over 10^7..10^8 terms is ~1e-6 of the sum; the weight-gradient side is fp32."""
    B = 64
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout + h)
    x = ops.to_nhwc(torch.rand(B, cin, h, w, generator=g).to(DEV), torch.bfloat16)
    dz = ops.to_nhwc(torch.rand(B, cout, h, w, generator=g).to(DEV), torch.bfloat16)
    wt = _bf(torch.rand(cout, cin, 3, 3, generator=g) / (9 * cin)).to(DEV)
    y = ops._conv(x, ops._pack(wt, False, torch.bfloat16), None, None, None, False, torch.bfloat16)
    dx = ops._conv(dz, ops._pack(wt, True, torch.bfloat16), None, None, None, False, torch.bfloat16)
    lib = L.load()
    dw, db = torch.empty(cout, cin, 3, 3, device=DEV), torch.empty(cout, device=DEV)
    need = lib.bx_conv3x3_wgrad_workspace(B, h, w, x.shape[3], cout, L.BX_BF16, L.BX_ALGO_AUTO)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    L.check(lib.bx_conv3x3_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, x.shape[3], cout, L.BX_BF16,
                                 L.BX_ALGO_AUTO, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "wgrad")
    a = float((y.double() * dz.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((wt.double() * dw.double()).sum())
    assert a > 0.5 * float(y.double().norm() * dz.double().norm())  # positive operands: the form is of the order of its bound
    # observed: the three agree to 1.4e-5 .. 3.1e-5 with a fixed sign (sums of same-sign terms in the MFMA's fp32 accumulator are
    # not unbiased); a tile pattern that drops or misplaces 1e-3 of the elements moves a side by ~1e-3
    print(f"[adjoint] {cin}->{cout} {h}x{w}: <y,dz>={a:.6e} <x,dx>/<y,dz>-1={b / a - 1:+.2e} <w,dw>/<y,dz>-1={c / a - 1:+.2e}")
    assert abs(a - c) <= 1e-4 * a, (a, c)
    assert abs(a - b) <= 1e-4 * a, (a, b)
    if h * w <= 512:                                                # anchor the late stages to an fp64 evaluation of the same form
        xd = ops.to_nchw_f32(x, cin).double()
        e = float((F.conv2d(xd, wt.double(), padding=1) * ops.to_nchw_f32(dz, cout).double()).sum())
        assert max(abs(a - e), abs(b - e), abs(c - e)) <= 1e-4 * e, (a, b, c, e)
    assert abs(float(db.double().sum()) - float(dz.double().sum())) <= 1e-5 * float(dz.double().abs().sum())


@pytest.mark.parametrize("conv_shape", [(64, 32, 16, 32), (32, 16, 40, 72), (256, 128, 8, 16), (16, 16, 64, 96)])
def test_conv_launch_carries_pending_weight_gradient_sum(conv_shape):
    """bx_conv3x3_carry: the pending partial sum of a chained weight gradient is finished by extra workgroups of a convolution
    launch (pixel-split and persistent kernels); the convolution's own result and the summed gradient are bit-identical to the
    separate launches, and nothing is left pending."""
    import ctypes
    lib = L.load()
    torch.manual_seed(9)
    B, cin, cout, h, w = 4, 64, 64, 16, 32
    xn = ops.to_nhwc(torch.randn(B, cin, h, w, device=DEV), torch.bfloat16)
    dzn = ops.to_nhwc(torch.randn(B, cout, h, w, device=DEV) * 0.1, torch.bfloat16)
    need = lib.bx_conv3x3_wgrad_workspace(B, h, w, cin, cout, L.BX_BF16, L.BX_ALGO_MFMA)
    want_dw, want_db = torch.empty(cout, cin, 3, 3, device=DEV), torch.empty(cout, device=DEV)
    ws0 = torch.empty(need, dtype=torch.uint8, device=DEV)
    L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), want_dw.data_ptr(), want_db.data_ptr(), B, h, w, cin, cin, cout, L.BX_BF16,
                                 L.BX_ALGO_MFMA, ws0.data_ptr(), ws0.numel(), 0), "wgrad")
    # the convolution that will carry the sum (a data-gradient-style call: mask + addend epilogue)
    ci2, co2, h2, w2 = conv_shape
    wt = _bf(torch.randn(ci2, co2, 3, 3) / (3 * ci2 ** 0.5)).to(DEV)       # layer co2 -> ci2; its data gradient maps ci2 -> co2
    packed = ops._pack(wt, True, torch.bfloat16)
    g2 = ops.to_nhwc(torch.randn(3, ci2, h2, w2, device=DEV), torch.bfloat16)
    mk = ops.to_nhwc(torch.randn(3, co2, h2, w2, device=DEV), torch.bfloat16)
    ad = ops.to_nhwc(torch.randn(3, co2, h2, w2, device=DEV), torch.bfloat16)
    y_plain = torch.empty(3, h2, w2, packed[3], dtype=torch.bfloat16, device=DEV)
    L.check(lib.bx_conv3x3(g2.data_ptr(), None, packed[1].data_ptr(), None, mk.data_ptr(), ad.data_ptr(), y_plain.data_ptr(), 3, h2, w2, g2.shape[3],
                           packed[3], L.BX_BF16, 0, L.BX_ALGO_MFMA, 0), "conv")
    pend = L.WgradPending()
    dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
    db = torch.full((cout,), float("nan"), device=DEV)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    L.check(lib.bx_conv3x3_wgrad_chained(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cin, cout, L.BX_BF16,
                                         L.BX_ALGO_MFMA, ws.data_ptr(), ws.numel(), ctypes.byref(pend), 0), "wgrad chained")
    assert pend.valid == 1
    torch.cuda.synchronize()
    assert torch.isnan(dw).all()
    y_carry = torch.empty_like(y_plain)
    L.check(lib.bx_conv3x3_carry(g2.data_ptr(), None, packed[1].data_ptr(), None, mk.data_ptr(), ad.data_ptr(), y_carry.data_ptr(), 3, h2, w2,
                                 g2.shape[3], packed[3], L.BX_BF16, 0, L.BX_ALGO_MFMA, ctypes.byref(pend), 0), "conv carry")
    assert pend.valid == 0
    torch.cuda.synchronize()
    assert torch.equal(y_carry, y_plain)
    assert rel_err(dw.cpu(), want_dw.cpu()) < 1e-6 and rel_err(db.cpu(), want_db.cpu()) < 1e-6
