"""GPU: fp32-STORAGE convolutions on the matrix cores (csrc/conv3x3_split.hip: operands split x = h + m + l, six bf16 MFMAs per
product) against (a) an fp64 CPU convolution of the same fp32 inputs -- the truth -- and (b) the VALU fp32 kernels they replace
(conv3x3.hip), whose own distance from the truth is printed beside the split kernels'.

Derived bound: the three bf16 terms carry fp32's 24 significand bits and the dropped partial products are <= 2^-26 of a product, so
what remains is fp32 accumulation: the matrix core adds 32 products per instruction into an fp32 accumulator, the VALU kernel one
fma at a time -- both land 1e-7 .. 1e-6 from the fp64 result relative to the output scale (2^-24 per addition, growing with
sqrt(K)).  Asserted: 2e-6, and within 4 x of the VALU kernel's own error (+1e-7).  (The two-term / three-MFMA form this file first
tested sat at 3-6e-6 whatever K: a sum of K products and its error are random walks of the same scale.)
Reference ops: nn.Conv2d(k=3, p=1) + F.relu and their autograd backward (models.py:49-51,64-66)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import brainxai
from brainxai import _lib as L
from brainxai import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 2e-6


def _err(got, want):
    want = want.double()
    return float((got.double() - want).abs().max() / want.abs().max())


SHAPES = [(4, 16, 9, 13), (16, 16, 8, 32), (16, 32, 17, 40), (32, 32, 16, 16), (32, 64, 8, 16), (64, 64, 5, 6), (64, 128, 16, 32),
          (128, 256, 8, 16), (256, 256, 4, 8), (128, 128, 25, 18)]


def _run_conv(xn, packed_f32, packed_split, bias, mask, addend, B, h, w, cip, cop, flags, algo):
    lib = L.load()
    y = torch.full((B, h, w, cop), float("nan"), dtype=torch.float32, device=DEV)
    L.check(lib.bx_conv3x3(xn.data_ptr(), packed_f32.data_ptr(), packed_split.data_ptr() if packed_split is not None else None,
                           bias.data_ptr() if bias is not None else None, mask.data_ptr() if mask is not None else None,
                           addend.data_ptr() if addend is not None else None, y.data_ptr(), B, h, w, cip, cop, L.BX_F32, flags, algo,
                           torch.cuda.current_stream().cuda_stream), "bx_conv3x3")
    return y


@pytest.mark.parametrize("cin,cout,h,w", SHAPES)
def test_conv_split_forward(cin, cout, h, w):
    torch.manual_seed(cin * 131 + cout)
    B = 3
    x = torch.randn(B, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    b = torch.randn(cout)
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    xn = ops.to_nhwc(x.to(DEV), torch.float32)
    pf = ops._pack(wt.to(DEV), False)[0]
    ps = ops._pack(wt.to(DEV), False, torch.float32)
    assert ps[1] is not None and ps[4] == "split", "the split operand must be produced for this shape"
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("split", L.BX_ALGO_MFMA), ("auto", L.BX_ALGO_AUTO)):
        y = _run_conv(xn, pf, ps[1], b.to(DEV), None, None, B, h, w, ops.pad8(cin), cout, L.BX_EPI_RELU, algo)
        outs[name] = ops.to_nchw_f32(y, cout).cpu()
    torch.cuda.synchronize()
    e_split, e_direct = _err(outs["split"], want), _err(outs["direct"], want)
    print(f"conv fwd {cin}->{cout} {h}x{w}: split {e_split:.2e}  VALU fp32 {e_direct:.2e}")
    assert not torch.isnan(outs["split"]).any()
    assert e_split < TOL and e_split < 4 * e_direct + 1e-7
    assert torch.equal(outs["auto"], outs["split"])         # AUTO takes the matrix-core path for fp32 storage when the operand is given


@pytest.mark.parametrize("cin,cout,h,w", SHAPES[1:])
def test_conv_split_data_gradient(cin, cout, h, w):
    """dX = conv(dZ, flipped W^T) * (y_below > 0) + addend: the epilogue operands are fp32 here (16-byte accesses)"""
    torch.manual_seed(cin * 17 + cout)
    B = 2
    dz = torch.randn(B, cout, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (3 * cout ** 0.5)
    ymask, add = torch.randn(B, cin, h, w), torch.randn(B, cin, h, w)
    want = F.conv_transpose2d(dz.double(), wt.double(), padding=1) * (ymask > 0) + add.double()
    dzn, mn, an = (ops.to_nhwc(t.to(DEV), torch.float32) for t in (dz, ymask, add))
    pf = ops._pack(wt.to(DEV), True)[0]
    ps = ops._pack(wt.to(DEV), True, torch.float32)
    assert ps[1] is not None
    outs = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("split", L.BX_ALGO_MFMA)):
        y = _run_conv(dzn, pf, ps[1], None, mn, an, B, h, w, cout, ops.pad8(cin), 0, algo)
        outs[name] = ops.to_nchw_f32(y, cin).cpu()
    torch.cuda.synchronize()
    e_split, e_direct = _err(outs["split"], want), _err(outs["direct"], want)
    print(f"conv dgrad {cout}->{cin} {h}x{w}: split {e_split:.2e}  VALU fp32 {e_direct:.2e}")
    assert e_split < TOL and e_split < 4 * e_direct + 1e-7


WG_SHAPES = [(4, 8, 16, 9, 13), (16, 16, 16, 8, 32), (16, 16, 32, 17, 40), (32, 32, 16, 24, 24), (32, 32, 32, 8, 16), (32, 32, 64, 16, 32),
             (64, 64, 64, 5, 6), (64, 64, 128, 16, 32), (128, 128, 256, 8, 16), (256, 256, 256, 4, 8), (128, 128, 128, 25, 18)]


@pytest.mark.parametrize("cin,cip,cout,h,w", WG_SHAPES)
def test_wgrad_split(cin, cip, cout, h, w):
    torch.manual_seed(cin + 3 * cout)
    B = 3
    x = torch.randn(B, cin, h, w)
    dz = torch.randn(B, cout, h, w)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wt, bias, padding=1) * dz.double()).sum().backward()
    xn = ops.to_nhwc(x.to(DEV), torch.float32)
    assert xn.shape[3] == cip
    dzn = ops.to_nhwc(dz.to(DEV), torch.float32)
    lib = L.load()
    got = {}
    for name, algo in (("direct", L.BX_ALGO_DIRECT), ("split", L.BX_ALGO_MFMA), ("auto", L.BX_ALGO_AUTO)):
        need = lib.bx_conv3x3_wgrad_workspace(B, h, w, cip, cout, L.BX_F32, algo)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=DEV)
        db = torch.full((cout,), float("nan"), device=DEV)
        L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cip, cout, L.BX_F32, algo,
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), name)
        got[name] = (dw.cpu(), db.cpu())
    torch.cuda.synchronize()
    e_split, e_direct = _err(got["split"][0], wt.grad), _err(got["direct"][0], wt.grad)
    print(f"wgrad {cin}->{cout} {h}x{w}: split {e_split:.2e}  VALU fp32 {e_direct:.2e}  bias {_err(got['split'][1], bias.grad):.2e}")
    assert not torch.isnan(got["split"][0]).any() and not torch.isnan(got["split"][1]).any()
    assert e_split < TOL and e_split < 4 * e_direct + 1e-7
    assert _err(got["split"][1], bias.grad) < 2e-6            # the bias gradient sums the fp32 values themselves
    assert torch.equal(got["auto"][0], got["split"][0]) and torch.equal(got["auto"][1], got["split"][1])


def test_wgrad_split_in_a_chain():
    """the chained entry point with fp32 storage: the split kernel does not carry another layer's sum, it finishes the pending one
    first and leaves its own pending; results equal the immediate form bit for bit"""
    lib = L.load()
    torch.manual_seed(5)
    B = 2
    layers = [(16, 32, 16, 32), (32, 32, 8, 16)]
    pend = L.WgradPending()
    keep, got, want = [], [], []
    for cin, cout, h, w in layers:
        xn = ops.to_nhwc(torch.randn(B, cin, h, w, device=DEV), torch.float32)
        dzn = ops.to_nhwc(torch.randn(B, cout, h, w, device=DEV), torch.float32)
        need = lib.bx_conv3x3_wgrad_workspace(B, h, w, cin, cout, L.BX_F32, L.BX_ALGO_AUTO)
        for chained in (False, True):
            ws = torch.empty(need, dtype=torch.uint8, device=DEV)
            dw, db = torch.full((cout, cin, 3, 3), float("nan"), device=DEV), torch.full((cout,), float("nan"), device=DEV)
            if chained:
                L.check(lib.bx_conv3x3_wgrad_chained(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cin, cout, L.BX_F32,
                                                     L.BX_ALGO_AUTO, ws.data_ptr(), ws.numel(), ctypes.byref(pend), 0), "chained")
                got.append((dw, db))
            else:
                L.check(lib.bx_conv3x3_wgrad(xn.data_ptr(), dzn.data_ptr(), dw.data_ptr(), db.data_ptr(), B, h, w, cin, cin, cout, L.BX_F32,
                                             L.BX_ALGO_AUTO, ws.data_ptr(), ws.numel(), 0), "immediate")
                want.append((dw, db))
            keep.append((ws, xn, dzn))
    assert pend.valid == 1
    L.check(lib.bx_conv3x3_wgrad_finish(ctypes.byref(pend), 0), "finish")
    torch.cuda.synchronize()
    assert pend.valid == 0
    for (gw, gb), (ww, wb) in zip(got, want):
        assert torch.equal(gw, ww) and torch.equal(gb, wb)


def test_pack_many_split_equals_single_packs():
    """ops.PackPlan(dtype=float32): all split operands of a model from ONE launch (job bit 1), bit-identical to bx_conv3x3_pack_split"""
    torch.manual_seed(9)
    ws = [torch.randn(co, ci, 3, 3, device=DEV) for ci, co in ((4, 16), (16, 16), (16, 32), (64, 128), (256, 256))]
    plan = ops.PackPlan(ws, torch.float32)
    plan.run()
    for i, w in enumerate(ws):
        for flip in (False, True):
            view = plan.get(i, flip)
            single = ops._pack(w, flip, torch.float32)
            if single[1] is None:
                assert view is None
                continue
            assert view is not None and view[4] == "split" and view[2:4] == single[2:4]
            assert torch.equal(view[1], single[1]), (i, flip)
    torch.cuda.synchronize()


def test_split_operand_is_h_plus_m_plus_l():
    """h + m + l reproduces the fp32 weight exactly (three 8-bit significands cover its 24) and h is the weight's bf16 rounding"""
    w = torch.randn(32, 16, 3, 3, device=DEV)
    pf, pm, ip, op, layout = ops._pack(w, False, torch.float32)
    assert pf is None and layout == "split"
    n = pm.numel() // 6
    imgs = pm.view(torch.bfloat16).float().view(3, n)
    bf = ops._pack(w, False, torch.bfloat16)[1].view(torch.bfloat16).float()
    assert torch.equal(imgs[0], bf)
    # element (chunk 0, K-step s, row o, kk): q = 32 s + kk, tap = q // 16, i = q % 16
    total = (imgs[0].double() + imgs[1].double() + imgs[2].double()).float()
    for s_, o_, kk in ((2, 5, 7), (0, 0, 0), (4, 31, 15)):
        q = 32 * s_ + kk
        tap, i = q // 16, q % 16
        idx = (s_ * op + o_) * 32 + kk
        assert float(total[idx]) == float(w[o_, i, tap // 3, tap % 3])
    torch.cuda.synchronize()


def test_fp32_block_runs_on_the_matrix_cores():
    """a compute_dtype=float32 Block goes through bx_conv3x3 with the split operand (no VALU convolution left in the fp32 path of the
    benchmark model); forward and every gradient agree with the VALU path to the split bound on a training-mode pass"""
    torch.manual_seed(3)
    blk = brainxai.Block(16, 32, "max", (2, 2), dropout_p=0.0).to(DEV)
    blk.train()
    x = torch.randn(4, 16, 32, 48, device=DEV)
    r = torch.randn(4, 32, 16, 24, device=DEV)
    res = {}
    try:
        for name, algo in (("mfma", L.BX_ALGO_AUTO), ("direct", L.BX_ALGO_DIRECT)):
            ops.CONV_ALGO = ops.WGRAD_ALGO = algo
            blk.zero_grad()
            xi = x.clone().requires_grad_(True)
            y = blk(xi)
            (y * r).sum().backward()
            res[name] = (y.detach().clone(), xi.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()})
    finally:
        ops.CONV_ALGO = ops.WGRAD_ALGO = L.BX_ALGO_AUTO
    torch.cuda.synchronize()
    assert _err(res["mfma"][0], res["direct"][0]) < 2e-6
    # gradients: a ReLU / max-pool decision with a 1e-7 margin may still differ between the two fp32-grade forwards; relative L2 is the robust measure
    dxa, dxd = res["mfma"][1].double(), res["direct"][1].double()
    assert float((dxa - dxd).norm() / dxd.norm()) < 1e-3
    for n in res["mfma"][2]:
        a, d = res["mfma"][2][n].double(), res["direct"][2][n].double()
        assert float((a - d).norm() / (d.norm() + 1e-30)) < 1e-3, n
