"""SURVEY 8f rank 4 (training step): the cald_train_* device operators against torch-CPU autograd in float64.

The reference's training arithmetic is torchvision 0.8.2 modules under torch autograd (cald_train.py:40-74); the checker here is
the same torch ops on the CPU in double precision, the tolerance is float32 accumulation noise (stated per test)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    from cald_amd import train_ops
    assert torch.cuda.is_available()
    return torch, train_ops


def _close(got, want, tol, what):
    got = got.detach().double().cpu().numpy(); want = want.detach().double().cpu().numpy()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = max(1e-30, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err <= tol, "%s: max err / max|ref| = %.3g > %.3g" % (what, err, tol)


CONV_CASES = [   # N, H, W, Cin, Cout, K, stride, pad, bias
    (2, 19, 23, 64, 128, 3, 1, 1, False),
    (3, 16, 20, 128, 64, 1, 1, 0, False),
    (2, 18, 22, 64, 64, 3, 2, 1, False),
    (2, 18, 22, 64, 128, 1, 2, 0, False),
    (1, 13, 17, 256, 256, 3, 1, 1, True),
    (2, 12, 10, 256, 15, 1, 1, 0, True),      # merged RPN head (3 logits + 12 deltas): dY rows padded to 16 channels
    (2, 9, 11, 48, 80, 3, 1, 1, True),        # channel counts that are not multiples of 64
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_forward_dgrad_wgrad_vs_autograd(T, case):
    """forward, data gradient (flipped-filter conv, strided layers through the dilated grid) and weight / bias gradient
    (split-K MFMA GEMM + fixed-order reduction) of one conv layer == torch.nn.functional.conv2d + autograd in float64,
    to 2e-5 of the tensor's largest magnitude (float32 sums over up to 2 304 x pixels terms)."""
    torch, ops = T
    import torch.nn.functional as F
    N, H, W, Cin, Cout, K, s, p, has_bias = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g) if has_bias else None
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    bd = b.double().requires_grad_() if has_bias else None
    y = F.conv2d(xd, wd, bd, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())

    xc = x.permute(0, 2, 3, 1).contiguous().cuda(); wc = w.cuda(); bc = b.cuda() if has_bias else None
    pk = ops.PackedConv(wc, bias=bc)
    out = ops.conv(xc, pk, stride=s, pad=p)
    _close(out.permute(0, 3, 1, 2), y, 2e-5, "forward")

    ld = ops.round_up(Cout, 4)
    gc = torch.zeros(N, y.shape[2], y.shape[3], ld, device="cuda")
    gc[..., :Cout] = gy.permute(0, 2, 3, 1).cuda()
    pkd = ops.PackedConv(wc, CinK=ld, mode=1)
    dx = ops.conv_dgrad(gc, pkd, H, W, s, p)
    _close(dx.permute(0, 3, 1, 2), xd.grad, 2e-5, "data gradient")

    dw = torch.empty_like(wc); db = torch.empty(Cout, device="cuda") if has_bias else None
    ops.conv_wgrad(xc, gc, Cin, Cout, K, K, s, p, dw, db)
    _close(dw, wd.grad, 2e-5, "weight gradient")
    if has_bias:
        _close(db, bd.grad, 2e-5, "bias gradient")
    dw2 = dw.clone()
    ops.conv_wgrad(xc, gc, Cin, Cout, K, K, s, p, dw2, None, accumulate=True)
    assert torch.equal(dw2, dw + dw), "accumulate adds the same deterministic sum"


def test_epilogue_bn_relu_residual_and_relu_backward(T):
    """conv -> FrozenBatchNorm scale/shift -> + residual -> ReLU in one launch, and its backward mask + scale."""
    torch, ops = T
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    N, H, W, Cin, Cout = 2, 14, 18, 64, 256
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 1, 1, generator=g) / 8
    sc = torch.rand(Cout, generator=g) + 0.5; sh = torch.randn(Cout, generator=g); res = torch.randn(N, Cout, H, W, generator=g)
    y = F.relu(F.conv2d(x.double(), w.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double())
    pk = ops.PackedConv(w.cuda(), scale=sc.cuda(), shift=sh.cuda())
    out = ops.conv(x.permute(0, 2, 3, 1).contiguous().cuda(), pk, relu=True, residual=res.permute(0, 2, 3, 1).contiguous().cuda())
    _close(out.permute(0, 3, 1, 2), y, 2e-5, "conv+bn+residual+relu")
    gy = torch.randn(N, H, W, Cout, generator=g).cuda()
    want = gy * (out > 0).float() * sc.cuda()
    got = ops.relu_bwd_(gy.clone(), out, sc.cuda())
    assert torch.equal(got, want)


def test_fc6_on_roi_rows_and_linear_gradients(T):
    """box_head.fc6 applied to RoIAlign rows laid out [R][7*7][256] with the torch weight [1024][256*7*7] (mode 2), its data
    gradient and weight gradient (written back in the torch layout), and a plain linear layer (predictor, 105 outputs)."""
    torch, ops = T
    g = torch.Generator().manual_seed(5)
    R, Cc, taps, Co = 200, 64, 49, 128
    feat = torch.randn(R, Cc, 7, 7, generator=g)                      # what torchvision flattens: [R, C*7*7]
    w = torch.randn(Co, Cc * taps, generator=g) / 56; b = torch.randn(Co, generator=g)
    xd = feat.double().flatten(1).requires_grad_(); wd = w.double().requires_grad_(); bd = b.double().requires_grad_()
    y = torch.relu(xd @ wd.t() + bd)
    gy = torch.randn(R, Co, generator=g)
    y.backward(gy.double())
    rows = feat.permute(0, 2, 3, 1).contiguous().cuda()               # [R, 7, 7, C] = [R][tap][C]
    pk = ops.PackedConv(w.cuda(), bias=b.cuda(), mode=2, taps=taps)
    out = ops.conv(rows.view(1, 1, R, taps * Cc), pk, relu=True).view(R, Co)
    _close(out, y, 2e-5, "fc6 forward")
    gz = ops.relu_bwd_(gy.cuda().clone(), out)
    dw = torch.empty_like(w, device="cuda"); db = torch.empty(Co, device="cuda")
    ops.linear_wgrad(rows.view(R, -1), gz, Co, dw, db, taps=taps)
    _close(dw, wd.grad, 2e-5, "fc6 weight gradient (torch layout)"); _close(db, bd.grad, 2e-5, "fc6 bias gradient")
    # data gradient wrt the rows: a plain linear layer with the tap-major weight, transposed
    wt = w.view(Co, Cc, taps).permute(0, 2, 1).reshape(Co, taps * Cc).contiguous().cuda()     # [Co][tap*C]: test-side reorder
    pkd = ops.PackedConv(wt, CinK=Co, mode=1)
    dx = ops.conv(gz.view(1, 1, R, Co), pkd).view(R, 7, 7, Cc)
    _close(dx.permute(0, 3, 1, 2).flatten(1), xd.grad, 2e-5, "fc6 data gradient")
    # predictor: 105 outputs, rows padded to 108
    w2 = torch.randn(105, Co, generator=g) / 11; b2 = torch.randn(105, generator=g)
    hd = y.detach().requires_grad_(); w2d = w2.double().requires_grad_(); b2d = b2.double().requires_grad_()
    z = hd @ w2d.t() + b2d
    gz2 = torch.randn(R, 105, generator=g); z.backward(gz2.double())
    pk2 = ops.PackedConv(w2.cuda(), bias=b2.cuda())
    zz = ops.conv(out.view(1, 1, R, Co), pk2, out_ld=108).view(R, 108)
    _close(zz[:, :105], z, 2e-5, "predictor forward"); assert float(zz[:, 105:].abs().max()) == 0.0
    gpad = torch.zeros(R, 108, device="cuda"); gpad[:, :105] = gz2.cuda()
    dw2 = torch.empty(105, Co, device="cuda"); db2 = torch.empty(105, device="cuda")
    ops.linear_wgrad(out, gpad, 105, dw2, db2)
    _close(dw2, w2d.grad, 2e-5, "predictor weight gradient"); _close(db2, b2d.grad, 2e-5, "predictor bias gradient")
    dh = ops.conv(gpad.view(1, 1, R, 108), ops.PackedConv(w2.cuda(), CinK=108, mode=1)).view(R, Co)
    _close(dh, hd.grad, 2e-5, "predictor data gradient")


def test_fpn_topdown_forward_and_backward_pieces(T):
    """lateral 1x1 conv + nearest-upsampled coarser level in one launch; upsample backward sums the 2x2 fine pixels."""
    torch, ops = T
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    N, Hc, Wc, Hf, Wf, Cc = 2, 7, 9, 14, 18, 256
    coarse = torch.randn(N, Cc, Hc, Wc, generator=g); xf = torch.randn(N, 128, Hf, Wf, generator=g)
    w = torch.randn(Cc, 128, 1, 1, generator=g) / 11; b = torch.randn(Cc, generator=g)
    want = F.conv2d(xf.double(), w.double(), b.double()) + F.interpolate(coarse.double(), size=(Hf, Wf), mode="nearest")
    pk = ops.PackedConv(w.cuda(), bias=b.cuda())
    got = ops.conv(xf.permute(0, 2, 3, 1).contiguous().cuda(), pk, up=coarse.permute(0, 2, 3, 1).contiguous().cuda())
    _close(got.permute(0, 3, 1, 2), want, 2e-5, "lateral + top-down")
    gf = torch.randn(N, Hf, Wf, Cc, generator=g); gc0 = torch.randn(N, Hc, Wc, Cc, generator=g)
    cd = coarse.double().requires_grad_()
    F.interpolate(cd, size=(Hf, Wf), mode="nearest").backward(gf.permute(0, 3, 1, 2).double())
    got = ops.upsample_bwd_(gf.cuda(), gc0.cuda().clone())
    _close(got.permute(0, 3, 1, 2), cd.grad + gc0.permute(0, 3, 1, 2).double(), 1e-6, "upsample backward")


def test_sgd_step_equals_torch_optim(T):
    torch, ops = T
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(10007, generator=g)
    ref = torch.nn.Parameter(p0.clone()); opt = torch.optim.SGD([ref], lr=0.0025, momentum=0.9, weight_decay=1e-4)
    p = p0.cuda(); buf = torch.zeros_like(p)
    for step in range(3):
        gr = torch.randn(10007, generator=g)
        ref.grad = gr.clone(); opt.step()
        ops.sgd_(p, gr.cuda(), buf, 0.0025, 0.9, 1e-4, step == 0)
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=0, atol=5e-7)     # one float32 ulp: torch folds -lr * buf into an fma
