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
    if not torch.cuda.is_available():      # module-scoped: runs before conftest's function-scoped auto-skip
        pytest.skip("needs an MI355X")
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
    (2, 18, 22, 128, 64, 3, 2, 1, False),     # Cin % 128 == 0: the weight gradient's offset-table variant, strided
    (2, 18, 22, 256, 64, 1, 2, 0, True),      # ... the 1x1 stride-2 downsample shape
    (2, 75, 70, 128, 128, 3, 1, 1, False),    # ... 10 500 pixels: several splits, a ragged last one
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


@pytest.mark.parametrize("hw", [(18, 22), (17, 23), (9, 8)])
def test_stride2_data_gradient_by_phases(T, hw):
    """The 3x3 stride-2 data gradient as four phase convolutions on the un-dilated dY + weave (even and odd input sizes), with the
    FrozenBN scale folded in and the next ReLU's mask applied in the weave == autograd."""
    torch, ops = T
    import torch.nn.functional as F
    H, W = hw
    g = torch.Generator().manual_seed(H * 31 + W)
    N, Cin, Cout = 2, 64, 128
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24; sc = torch.rand(Cout, generator=g) + 0.5
    act = torch.randn(N, Cin, H, W, generator=g)
    xd = x.double().requires_grad_()
    y = F.conv2d(xd, w.double(), stride=2, padding=1) * sc.double().view(1, -1, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    packs = ops.pack_s2_grads(w.cuda(), scale=sc.cuda(), CinK=Cout)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    got = ops.conv_dgrad_s2(nhwc(gy), packs, H, W, mask=nhwc(act))
    _close(got.permute(0, 3, 1, 2), xd.grad * (act.double() > 0), 2e-5, "stride-2 data gradient by phases")
    got = ops.conv_dgrad_s2(nhwc(gy), packs, H, W)
    _close(got.permute(0, 3, 1, 2), xd.grad, 2e-5, "stride-2 data gradient by phases, no mask")


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


def test_data_gradient_with_fused_relu_backward(T):
    """The data gradient with the NEXT ReLU's backward fused into its epilogue (out = act > 0 ? out : 0), on the tiled kernel
    (full and ragged tiles), in a grouped launch, and through the separate pass taken for shapes the tiled kernel does not cover."""
    torch, ops = T
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(13)
    for (N, H, W, Cin, Cout, K) in ((2, 19, 23, 64, 128, 3), (1, 16, 16, 256, 64, 1), (2, 9, 11, 48, 80, 3)):
        w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
        gy = torch.randn(N, Cout, H, W, generator=g); act = torch.randn(N, Cin, H, W, generator=g); res = torch.randn(N, Cin, H, W, generator=g)
        want = (F.conv_transpose2d(gy.double(), w.double(), padding=K // 2) + res.double()) * (act.double() > 0)
        pkd = ops.PackedConv(w.cuda(), CinK=Cout, mode=1)
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
        got = ops.conv_dgrad(nhwc(gy), pkd, H, W, 1, K // 2, residual=nhwc(res), mask=nhwc(act))
        _close(got.permute(0, 3, 1, 2), want, 2e-5, "dgrad + residual + mask %s" % ((N, H, W, Cin, Cout, K),))
    w = torch.randn(256, 256, 3, 3, generator=g) / 48
    pkd = ops.PackedConv(w.cuda(), CinK=256, mode=1)
    gys = [torch.randn(2, 256, h, ww, generator=g) for h, ww in ((12, 14), (6, 7), (3, 4))]
    acts = [torch.randn(2, 256, h, ww, generator=g) for h, ww in ((12, 14), (6, 7), (3, 4))]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    outs = ops.conv_group([nhwc(t) for t in gys], pkd, pad=1, masks=[nhwc(t) for t in acts])
    for o, gy, act in zip(outs, gys, acts):
        _close(o.permute(0, 3, 1, 2), F.conv_transpose2d(gy.double(), w.double(), padding=1) * (act.double() > 0), 2e-5, "grouped dgrad + mask")


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


def test_training_proposals_equal_the_inference_oracle(T, oracle):
    """RegionProposalNetwork.filter_proposals at TRAINING sizes (2 000 per level before NMS, 2 000 after): the HIP kernels at
    the larger capacity == the C oracle's rpn_proposals on the same head tensors, bit for bit (index work)."""
    torch, ops = T
    g = torch.Generator().manual_seed(11)
    N, Hp, Wp = 2, 256, 320
    level_hw = [(64, 80), (32, 40), (16, 20), (8, 10), (4, 5)]
    heads = []
    for h, w in level_hw:
        t = torch.zeros(N, h, w, 16)
        t[..., :3] = torch.randn(N, h, w, 3, generator=g) * 2
        t[..., 3:15] = torch.randn(N, h, w, 12, generator=g) * 0.3
        heads.append(t)
    sizes = [(250, 300), (256, 317)]
    props, counts = ops.rpn_proposals([h.cuda() for h in heads], Hp, Wp, sizes, 2000, 2000, 0.7, 1e-3)
    base = np.concatenate([oracle.base_anchors([32.0 * 2 ** l], [0.5, 1.0, 2.0]) for l in range(5)])
    for i in range(N):
        want, _ = oracle.rpn_proposals([h[i].numpy() for h in heads], base, Hp, Wp, sizes[i][0], sizes[i][1], 3, 2000, 2000, 0.7, 1e-3)
        n = int(counts[i])
        assert n == want.shape[0] and n > 1000
        np.testing.assert_array_equal(props[i, :n].cpu().numpy(), want)


def test_anchors_matcher_and_box_coder(T, oracle):
    torch, ops = T
    from oracle import torch_train as tt
    g = torch.Generator().manual_seed(2)
    level_hw = [(48, 64), (24, 32), (12, 16), (6, 8), (3, 4)]
    anchors = ops.anchors(192, 256, level_hw, torch.device("cuda"))
    want = []
    for l, (h, w) in enumerate(level_hw):
        base = torch.from_numpy(oracle.base_anchors([32.0 * 2 ** l], [0.5, 1.0, 2.0])).reshape(-1, 4)
        ys, xs = torch.meshgrid(torch.arange(h) * (192 // h), torch.arange(w) * (256 // w), indexing="ij")
        want.append((torch.stack([xs, ys, xs, ys], dim=-1).reshape(-1, 1, 4).float() + base[None]).reshape(-1, 4))
    want = torch.cat(want)
    assert torch.equal(anchors.cpu(), want)
    xy = torch.rand(7, 2, generator=g) * torch.tensor([200.0, 140.0]); wh = torch.rand(7, 2, generator=g) * 90 + 8
    gt = torch.cat([xy, xy + wh], dim=1)
    for hi, lo, low in ((0.7, 0.3, True), (0.5, 0.5, False)):
        got = ops.match(anchors, gt.cuda(), hi, lo, low).cpu().long()
        assert torch.equal(got, tt.matcher(tt.box_iou(gt, want), hi, lo, low))
    pos = torch.nonzero(got >= 0).squeeze(1)
    assert len(pos) > 3
    enc = ops.box_encode(gt[got[pos]].cuda().contiguous(), anchors[pos.cuda()].contiguous(), (10.0, 10.0, 5.0, 5.0))
    _close(enc, tt.encode(gt[got[pos]].double(), want[pos].double(), (10.0, 10.0, 5.0, 5.0)), 1e-6, "BoxCoder.encode")


def test_roi_align_forward_backward_vs_autograd(T, oracle):
    torch, ops = T
    from oracle import torch_train as tt
    g = torch.Generator().manual_seed(4)
    N, Cc = 2, 16
    dims = [(40, 56), (20, 28), (10, 14), (5, 7)]
    feats = [torch.randn(N, Cc, h, w, generator=g) for h, w in dims]
    boxes = []
    for s in (12, 30, 70, 150, 400, 700):             # square roots of the areas that land on every pyramid level
        for _ in range(3):
            x, y = float(torch.rand(1, generator=g)) * 120 - 10, float(torch.rand(1, generator=g)) * 90 - 10
            boxes.append([x, y, x + s * 1.3, y + s / 1.3])
    boxes = torch.tensor(boxes); img = torch.arange(len(boxes)) % N
    fd = [f.double().requires_grad_() for f in feats]
    want = tt.roi_align(fd, img, boxes)                # [R, C, 7, 7]
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy.double())
    fc = [f.permute(0, 2, 3, 1).contiguous().cuda() for f in feats]
    rois = torch.cat([img[:, None].float(), boxes], dim=1).cuda().contiguous()
    got = ops.roi_align(fc, rois)                      # [R, 49, C]
    _close(got.view(-1, 7, 7, Cc).permute(0, 3, 1, 2), want, 1e-5, "RoIAlign forward")
    gf = [torch.zeros_like(f) for f in fc]
    ops.roi_align_bwd_(gf, rois, gy.permute(0, 2, 3, 1).reshape(-1, 49, Cc).contiguous().cuda())
    for l in range(4):
        _close(gf[l].permute(0, 3, 1, 2), fd[l].grad, 1e-5, "RoIAlign backward level %d" % l)


def test_losses_value_and_gradient(T, oracle):
    torch, ops = T
    import torch.nn.functional as F
    from oracle import torch_train as tt
    g = torch.Generator().manual_seed(6)
    R, Cc, ld = 300, 21, 108
    z = torch.zeros(R, ld); z[:, :105] = torch.randn(R, 105, generator=g) * 2
    lab = torch.randint(0, Cc, (R,), generator=g)
    zd = z.double().requires_grad_()
    ce = F.cross_entropy(zd[:, :Cc], lab)
    pos = torch.nonzero(lab > 0).squeeze(1); tgt = torch.randn(len(pos), 4, generator=g)
    sl = tt.smooth_l1_sum(zd[:, Cc:105].reshape(R, -1, 4)[pos, lab[pos]], tgt.double(), 1.0 / 9) / R
    (0.7 * ce + 1.3 * sl).backward()
    zc = z.cuda(); grad = torch.zeros_like(zc)
    l1 = ops.softmax_ce(zc, lab.cuda(), Cc, grad=grad, gscale=0.7)
    idx = (pos * ld + Cc + 4 * lab[pos]).cuda()
    l2 = ops.smooth_l1(zc, idx, tgt.cuda(), 1.0 / 9, R, grad=grad, gscale=1.3)
    _close(l1, ce.reshape(1), 1e-6, "cross entropy"); _close(l2, sl.reshape(1), 1e-6, "smooth L1")
    _close(grad, zd.grad, 1e-5, "d(0.7 ce + 1.3 smooth_l1)/d logits")
    x = torch.randn(5000, generator=g) * 3; sel = torch.randperm(5000, generator=g)[:256]; y = (torch.rand(256, generator=g) < 0.5).float()
    xd = x.double().requires_grad_()
    bce = F.binary_cross_entropy_with_logits(xd[sel], y.double()); bce.backward()
    gx = torch.zeros(5000, device="cuda")
    l3 = ops.bce_logits(x.cuda(), sel.cuda(), y.cuda(), grad=gx)
    _close(l3, bce.reshape(1), 1e-6, "BCE with logits"); _close(gx, xd.grad, 1e-5, "BCE gradient")


def _train_case(torch, n_images=2, seed=0, scale=0.4):
    from cald_amd import synth
    sd = synth.pseudo_trained_frcnn(21, 50, seed=3)
    imgs = synth.make_pool(n_images, "voc", seed, scale=scale)                   # scale 0.4: ~150 x 200 uint8 HWC
    images = [torch.from_numpy(im).permute(2, 0, 1).float().div(255) for im in imgs]
    rs = np.random.RandomState(seed + 1)
    targets = []
    for im in imgs:
        H, W = im.shape[:2]
        k = 2 + rs.randint(0, 3)
        x0 = rs.rand(k) * W * 0.6; y0 = rs.rand(k) * H * 0.6
        bw = W * (0.15 + 0.3 * rs.rand(k)); bh = H * (0.15 + 0.3 * rs.rand(k))
        boxes = np.stack([x0, y0, np.minimum(x0 + bw, W - 1), np.minimum(y0 + bh, H - 1)], axis=1).astype(np.float32)
        targets.append({"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(rs.randint(1, 21, k).astype(np.int64))})
    return sd, images, targets


def test_training_step_losses_and_gradients_vs_autograd(T, oracle):
    """One whole training step (forward in train mode, the four losses, the gradient of their sum wrt EVERY trainable tensor:
    layers 2-4, FPN, RPN head, box head, predictor; 64 sampled RoIs per image to keep the CPU checker quick) against torch-CPU autograd in float64 on the same images, targets,
    proposals, sampler draws (validated by the oracle as legal draws) and ReLU decisions.  Losses to 1e-4 relative; each gradient tensor to 1e-4 of its largest
    magnitude (float32 forward + backward vs float64)."""
    torch, ops = T
    from cald_amd import train
    from oracle import torch_train as tt
    sd, images, targets = _train_case(torch)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(7))
    losses = net.forward(images, targets)
    props = [p.cpu() for p in net.last["proposals"]]
    assert all(p.shape[0] > 100 for p in props)
    grads = {k: v.clone() for k, v in net.backward().items()}
    ref = tt.TorchTrainFRCNN(sd, 21, min_size=160, max_size=256)
    ref.masks = net.relu_decisions()           # both sides take the same branch at every differentiated ReLU (see TorchTrainFRCNN.relu)
    want, rec = ref.losses(images, targets, props, None, cfg=dict(box_batch=64), samples=net.last["samples"])
    assert torch.equal(rec["roi_labels"], net.last["roi_labels"]), "same sampled RoIs"
    for k in want:
        got = float(losses[k]); w = float(want[k].detach())
        assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (k, got, w)
    sum(want.values()).backward()
    tr = ref.trainable()
    assert sorted(tr) == sorted(grads), "same set of trainable tensors as resnet_fpn_backbone(trainable_layers=3)"
    worst = ("", 0.0)
    for k, g in grads.items():
        w = tr[k].grad
        assert w is not None, k
        scale = float(w.abs().max())
        assert scale > 0, k
        err = float((g.double().cpu() - w).abs().max()) / scale
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 1e-4, "largest gradient error %.3g at %s" % (worst[1], worst[0])


def test_drop_in_training_loop_updates_like_torch_sgd(T, oracle):
    """The reference's loop body verbatim (cald_train.py:54-71): loss_dict = model(images, targets); losses = sum(...);
    optimizer.zero_grad(); losses.backward(); optimizer.step(); lr_scheduler.step() -- with the HIP model, the HIP SGD and
    torch's own LambdaLR warmup.  After two iterations the parameters equal float64 autograd + torch.optim.SGD to 1e-5 of each
    tensor's largest magnitude."""
    torch, ops = T
    from cald_amd import train
    from oracle import torch_train as tt
    sd, images, targets = _train_case(torch, seed=5)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(1))
    model = train.TrainableFasterRCNN(net)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = train.SGD(params, lr=0.002, momentum=0.9, weight_decay=1e-4, net=net)
    warm = lambda x: 1.0 if x >= 3 else 0.001 * (1 - x / 3.0) + x / 3.0        # utils.warmup_lr_scheduler
    sched = torch.optim.lr_scheduler.LambdaLR(opt, warm)
    ref = tt.TorchTrainFRCNN(sd, 21, min_size=160, max_size=256)
    rparams = ref.trainable()
    ropt = torch.optim.SGD(list(rparams.values()), lr=0.002, momentum=0.9, weight_decay=1e-4)
    rsched = torch.optim.lr_scheduler.LambdaLR(ropt, warm)
    seen = []
    for it in range(2):
        loss_dict = model(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        seen.append(float(losses.detach()))
        opt.zero_grad(); losses.backward(); opt.step(); sched.step()
        ref.masks = net.relu_decisions()
        want, _ = ref.losses(images, targets, [p.cpu() for p in net.last["proposals"]], None, cfg=dict(box_batch=64), samples=net.last["samples"])
        rl = sum(want.values())
        assert abs(float(rl) - seen[-1]) <= 2e-4 * max(1.0, abs(float(rl))), (it, float(rl), seen[-1])
        ropt.zero_grad(); rl.backward(); ropt.step(); rsched.step()
    for k, p in net.named_parameters():
        w = rparams[k].detach()
        err = float((p.detach().double().cpu() - w).abs().max()) / float(w.abs().max())
        assert err <= 1e-5, (k, err)


def test_speculative_rpn_branch_equals_the_plain_backward(T):
    """The RPN branch of the backward pass is enqueued during the forward's RoI-sampling window for unit upstream gradients.  Same
    losses and the same gradients, bit for bit, as with the speculation off; non-unit upstream gradients fall back to the plain path;
    a forward under torch.no_grad() does not speculate."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, n_images=3, seed=33)
    out = {}
    for spec in (True, False):
        net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(4))
        net.speculate = spec
        losses = net.forward(images, targets)
        assert (net.last["spec"] is not None) == spec
        grads = net.backward()
        torch.cuda.synchronize()
        out[spec] = ({k: float(v) for k, v in losses.items()}, {k: v.clone() for k, v in grads.items()})
        if spec:                                            # scaled losses: the speculative result is dropped, the plain path runs
            net.forward(images, targets)
            g2 = net.backward((1.0, 1.0, 0.5, 2.0))
            torch.cuda.synchronize()
            k = "rpn.head.conv.weight"
            assert not torch.equal(g2[k], out[True][1][k]) and bool(torch.isfinite(g2[k]).all())
            model = train.TrainableFasterRCNN(net)
            with torch.no_grad():
                model(images, targets)
            assert net.last["spec"] is None
    assert out[True][0] == out[False][0]
    for k in out[True][1]:
        assert torch.equal(out[True][1][k], out[False][1][k]), k


def test_fused_sgd_launch_equals_the_per_tensor_launches(T):
    """train.SGD over exactly the trainer's parameters updates the flat buffers with one launch; with a parameter subset (or net=None)
    it launches per tensor.  Both give the same bits after two steps, and state[p]['momentum_buffer'] is readable either way."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, seed=9)
    flats, moms = [], []
    for fused in (True, False):
        net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(2))
        model = train.TrainableFasterRCNN(net)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = train.SGD(params, lr=0.001, momentum=0.9, weight_decay=1e-4, net=net if fused else None)
        for it in range(2):
            losses = sum(model(images, targets).values())
            opt.zero_grad(); losses.backward(); opt.step()
        assert (opt._mflat is not None) == fused
        torch.cuda.synchronize()
        flats.append(net.flat.clone())
        moms.append([opt.state[p]["momentum_buffer"].clone() for p in params])
        assert all(m.shape == p.shape for m, p in zip(moms[-1], params))
    assert torch.equal(flats[0], flats[1])
    assert all(torch.equal(a, b) for a, b in zip(*moms))


@pytest.mark.parametrize("trainable_layers", [0, 4])
def test_other_trainable_layer_settings_vs_autograd(T, oracle, trainable_layers):
    """resnet_fpn_backbone(trainable_layers=0 / 4): body frozen entirely / layer 1 trained too -- the set of trainable tensors and
    every gradient still equal float64 autograd (the data gradient stops exactly where the frozen part begins)."""
    torch, ops = T
    from cald_amd import train
    from oracle import torch_train as tt
    sd, images, targets = _train_case(torch, n_images=1, seed=30 + trainable_layers)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=32, trainable_layers=trainable_layers, generator=torch.Generator().manual_seed(3))
    losses = net.forward(images, targets)
    grads = {k: v.clone() for k, v in net.backward().items()}
    ref = tt.TorchTrainFRCNN(sd, 21, min_size=160, max_size=256, trainable_layers=trainable_layers)
    ref.masks = net.relu_decisions()
    want, _ = ref.losses(images, targets, [p.cpu() for p in net.last["proposals"]], None, cfg=dict(box_batch=32), samples=net.last["samples"])
    sum(want.values()).backward()
    tr = ref.trainable()
    assert sorted(tr) == sorted(grads) and len(grads) == {0: 30, 4: 82}[trainable_layers]
    for k, g in grads.items():
        w = tr[k].grad
        assert float((g.double().cpu() - w).abs().max()) <= 1e-4 * float(w.abs().max()), k
    with pytest.raises(NotImplementedError):
        train.FasterRCNNTrainer(sd, 21, trainable_layers=5)


def test_gradient_accumulation_follows_autograd_semantics(T):
    """backward() twice without zero_grad() adds (here: exactly doubles, same batch and samples); zero_grad() starts over."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, n_images=2, seed=17)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator())
    model = train.TrainableDetector(net)
    params = dict(net.named_parameters())
    def step():
        net.generator.manual_seed(9)
        sum(model(images, targets).values()).backward()
    step()
    once = {k: p.grad.clone() for k, p in params.items()}
    step()
    for k, p in params.items():          # (tensors shared by five pyramid levels add level by level: equal to float32 rounding, not bitwise)
        assert float((p.grad - 2 * once[k]).abs().max()) <= 1e-5 * float(once[k].abs().max()), k
    for p in params.values():
        p.grad = None
    step()
    for k, p in params.items():
        assert torch.equal(p.grad, once[k]), k


def test_training_step_is_bit_reproducible(T):
    """Same weights, images, targets and sampler seed twice: identical losses AND identical gradients of every tensor, bit for bit --
    with the weight gradients on the side stream and RoIAlign backward scattering through atomics (fixed-point accumulation)."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, n_images=3, seed=21)
    runs = []
    for _ in range(2):
        net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator().manual_seed(5))
        losses = net.forward(images, targets)
        grads = net.backward()
        torch.cuda.synchronize()
        runs.append(({k: float(v) for k, v in losses.items()}, {k: v.clone() for k, v in grads.items()}))
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_pack_plan_writes_the_per_layer_packs_bit_for_bit(T):
    """cald_train_pack_plan_run (all layers, two launches) against one cald_train_pack_conv per layer and form: every packed buffer
    of a Faster R-CNN trainer byte-equal (forward and data-gradient forms, FrozenBN scale folded, bias vectors, fc6's tap-major
    modes 2 / 3, the 15- and 105-column merged heads), again after the weights changed."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, n_images=2, seed=3)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator().manual_seed(5))
    convs = [cv for cv in net.convs if cv.trainable]
    assert len(convs) > 40
    planned = [pk for cv in convs for pk in cv._plan_packs()]
    assert {pk.mode for pk in planned} == {0, 1, 2, 3}
    plan = ops.PackPlan(planned, net.dev)
    for turn in range(2):
        if turn:
            net.flat.mul_(1.25).add_(0.01)
        for pk in planned:
            pk.buf.fill_(float("nan"))
        plan.run()
        torch.cuda.synchronize()
        for cv in convs:
            for pk, (b, sc, sh) in zip(cv._plan_packs(), ((cv.b, cv.scale, cv.shift), (None, cv.scale, None))):
                ref = ops.PackedConv(cv.w, b, sc, sh, CinK=pk.CinK, mode=pk.mode, taps=cv.taps)
                torch.cuda.synchronize()
                n = ref.buf.numel()
                if pk.mode in (1, 3):           # the data-gradient forms carry no epilogue vectors (never read: flags 0)
                    n = ref.buf.numel() - _vec_floats(ops, pk)
                assert torch.equal(pk.buf[:n].view(torch.int32), ref.buf[:n].view(torch.int32)), (cv.w.shape, pk.mode, turn)
    # and the model: a step with the plan equals a step with per-layer packs
    res = []
    for use_plan in (True, False):
        train._PACK_PLAN = use_plan
        try:
            net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator().manual_seed(5))
            out = []
            for _ in range(2):
                losses = net.forward(images, targets); grads = net.backward(); torch.cuda.synchronize()
                out.append(({k: float(v) for k, v in losses.items()}, {k: v.clone() for k, v in grads.items()}))
                net.flat.add_(net.gflat, alpha=-1e-3)
            res.append(out)
            assert (net._pack_plan is not None) == use_plan
        finally:
            train._PACK_PLAN = True
    for a, b in zip(*res):
        assert a[0] == b[0]
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), k


def test_roi_gather_equals_the_torch_gathers_and_box_encode(T):
    """cald_train_roi_gather (one kernel behind the RoI sampler: RoIAlign's [R, 5] rows and the foreground rows' BoxCoder.encode) against the
    eight torch ops + cald_train_box_encode it replaced, on the sampler's own output block: identical bits."""
    torch, ops = T
    rs = np.random.RandomState(11)
    N, batch, post = 3, 64, 200
    n_gt = [3, 0, 2]
    slots = [post] * N
    rows = [post + g for g in n_gt]
    T_ = sum(rows)
    xy = rs.rand(T_, 2).astype(np.float32) * 300
    table = torch.from_numpy(np.concatenate([xy, xy + 5 + rs.rand(T_, 2).astype(np.float32) * 120], axis=1)).cuda()
    g_xy = rs.rand(sum(n_gt) + 1, 2).astype(np.float32) * 300
    gts_all = torch.from_numpy(np.concatenate([g_xy, g_xy + 20 + rs.rand(len(g_xy), 2).astype(np.float32) * 100], axis=1)).cuda()
    gts_all[-1] = 0
    matched = np.concatenate([np.where(rs.rand(r) < 0.25, rs.randint(0, max(g, 1), r), -1).astype(np.int32) if g else np.full(r, -1, np.int32)
                              for r, g in zip(rows, n_gt)] + [np.array([post, post - 7, post - 50], np.int32)])
    gt_labels = np.concatenate([rs.randint(1, 21, g).astype(np.int64) for g in n_gt] + [np.zeros(1, np.int64)])
    blk, cap, R, n_pos, per = ops.roi_sample_host(slots, n_gt, matched[T_:], matched, gt_labels, rs.rand(T_), batch, 0.25, 108, 21)
    assert R > 0 and n_pos > 0 and cap == N * batch
    dev = torch.from_numpy(blk).cuda()
    w = (10.0, 10.0, 5.0, 5.0)
    rois, tgt = ops.roi_gather(table, gts_all, dev, cap, R, n_pos, w)
    keep, gsel, pos = dev[:R], dev[cap:cap + R], dev[4 * cap:4 * cap + n_pos]
    img = torch.from_numpy(blk[5 * cap:].view(np.float32)[:R].copy()).cuda()
    boxes = table[keep]
    want_rois = torch.cat([img[:, None], boxes], dim=1)
    want_tgt = ops.box_encode(gts_all[gsel].contiguous(), boxes.contiguous(), w)[pos]
    torch.cuda.synchronize()
    assert torch.equal(rois, want_rois)
    assert torch.equal(tgt.view(torch.int32), want_tgt.contiguous().view(torch.int32))


def _vec_floats(ops, pk):
    """floats of the three epilogue vectors at the end of a packed buffer"""
    two_kn = ops.packed_floats(pk.Cout, pk.Cin, pk.KH, pk.KW, pk.CinK, pk.mode)
    # floats = 2 * Kpad * NPad + 3 * NPad, NPad = cout_pad(columns) (csrc/common.h)
    n_true = pk.Cin if pk.mode == 1 else (pk.Cin * pk.KH if pk.mode == 3 else pk.Cout)
    npad = (n_true + 127) // 128 * 128 if n_true >= 128 else ((n_true + 63) // 64 * 64 if n_true >= 64 else (n_true + 31) // 32 * 32)
    assert (two_kn - 3 * npad) % (2 * npad) == 0
    return 3 * npad


def test_stock_torch_optimizer_drives_the_hip_model(T):
    """cald_train.py:397 verbatim -- torch.optim.SGD over task_model.parameters(): the parameters are ordinary torch Parameters whose
    .grad the hand-written backward fills, so torch's own optimizer (and anything else that reads .grad) works; the result equals
    the HIP SGD kernel's to float32 rounding."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, seed=12)
    outs = []
    for use_torch in (True, False):
        net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(4))
        model = train.TrainableDetector(net)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = (torch.optim.SGD(params, lr=1e-4, momentum=0.9, weight_decay=1e-4) if use_torch
               else train.SGD(params, lr=1e-4, momentum=0.9, weight_decay=1e-4))
        snap = []
        for it in range(2):
            losses = sum(loss for loss in model(images, targets).values())
            opt.zero_grad(); losses.backward(); opt.step()
            snap.append({k: p.detach().clone() for k, p in net.named_parameters()})
        outs.append(snap)
    # after one step the two differ by the optimizers' float32 rounding only; the second forward then runs on parameters that differ in the
    # last bits, where a discrete decision (a proposal on an NMS / matcher threshold, hence another sampled RoI) may go the other way
    for it, tol in ((0, 2e-5), (1, 5e-3)):
        for k in outs[0][it]:
            scale = float(outs[1][it][k].abs().max())
            assert float((outs[0][it][k] - outs[1][it][k]).abs().max()) <= tol * scale, (it, k)


def test_fitting_one_batch_lowers_the_loss(T):
    """Ten plain SGD steps on one batch (same sampler permutations every step): the summed loss falls."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch, seed=9)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator())
    model = train.TrainableFasterRCNN(net)
    opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=5e-5, momentum=0.0, weight_decay=0.0, net=net)
    seen = []
    for it in range(10):
        net.generator.manual_seed(3)
        loss_dict = model(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        assert bool(torch.isfinite(losses))
        seen.append(float(losses.detach()))
        opt.zero_grad(); losses.backward(); opt.step()
    assert np.mean(seen[-3:]) < seen[0] - 0.05 and max(seen) <= seen[0] + 0.05, seen


def test_active_learning_cycle_train_then_sweep_with_the_same_model_object(T, oracle):
    """cald_train.py's cycle on ONE model object: task_model.train() -> train_one_epoch (engine mirror, HIP SGD, warmup) ->
    task_model.eval() -> get_uncertainty.  After training, the inference engine must run on the UPDATED weights: its sweep
    equals the C oracle prepared from model.state_dict(), bit for bit, and differs from the sweep before training."""
    torch, ops = T
    from cald_amd import detector, engine, synth, sweep, train
    sd, images, targets = _train_case(torch, n_images=4, seed=2)
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=160, max_size=256).to("cuda")
    model.load_state_dict(sd)
    pool = synth.make_pool(3, "voc", 7, scale=0.4)
    loader = [((torch.from_numpy(im),), (None,)) for im in pool]
    augs = ["flip", "smaller_resize"]
    before, _ = sweep.get_uncertainty(model, loader, augs, 21)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    assert len(params) == 72
    opt = train.SGD(params, lr=2e-4, momentum=0.9, weight_decay=1e-4)
    batches = [(tuple(images[:2]), tuple(targets[:2])), (tuple(images[2:]), tuple(targets[2:]))]
    history = engine.train_one_epoch(model, opt, batches, "cuda", cycle=0, epoch=0, print_freq=0)
    assert len(history) == 2 and all(np.isfinite(history))
    with pytest.raises(ValueError):
        model(list(images[:1]))                       # training mode needs targets (frcnn_la.py:247-248)
    model.eval()
    after, _ = sweep.get_uncertainty(model, loader, augs, 21)
    new_sd = model.state_dict()
    assert all(isinstance(v, torch.Tensor) and v.dtype == torch.float32 and not v.is_cuda for v in new_sd.values())     # as nn.Module.state_dict()
    assert not np.array_equal(new_sd["roi_heads.box_head.fc7.weight"].numpy(), np.asarray(sd["roi_heads.box_head.fc7.weight"]))
    np.testing.assert_array_equal(new_sd["backbone.body.layer1.0.conv1.weight"].numpy(), np.asarray(sd["backbone.body.layer1.0.conv1.weight"]))
    P = oracle.prepare_frcnn(new_sd, 21, 50)
    want, _ = oracle.get_uncertainty(P, pool, augs, 21, min_size=160, max_size=256)
    assert after == want
    assert after != before


def test_training_epoch_from_a_vocdevkit_directory_on_resident_batches(T, tmp_path):
    """The training input side end to end (cald_train.py:288-336 without torchvision): VOCdevkit tree -> voc_utils dataset ->
    aspect-ratio groups from the JPEG headers -> GroupedBatchSampler over SubsetRandomSampler(labeled_set) -> ResidentTrainLoader
    (JPEGs decoded once on the GPU, flip on the device) -> train_one_epoch.  Every batch holds one orientation (so it pads to the
    tight size), the pool images equal Pillow's decode, flipped images carry mirrored boxes inside the image, the losses are finite
    and the weights move."""
    import random
    from PIL import Image
    from torch.utils.data.sampler import SubsetRandomSampler
    torch, ops = T
    from cald_amd import detector, engine, synth, train, voc_utils as vu
    from cald_amd.group_by_aspect_ratio import GroupedBatchSampler, create_aspect_ratio_groups
    base = tmp_path / "VOCdevkit" / "VOC2007"
    for d in ("ImageSets/Main", "Annotations", "JPEGImages"):
        (base / d).mkdir(parents=True)
    sizes = [(120, 160), (160, 120), (120, 160), (160, 120), (120, 160), (160, 120), (120, 160), (160, 120), (110, 160)]
    stems = ["%06d" % (7 * i + 3) for i in range(len(sizes))]
    for i, ((H, W), stem) in enumerate(zip(sizes, stems)):
        Image.fromarray(synth.synth_image(i, H, W)).save(str(base / "JPEGImages" / (stem + ".jpg")), quality=90)
        objs = "".join("<object><name>%s</name><difficult>0</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox></object>"
                       % (vu.VOC_CLASSES[1 + (i + j) % 20], 5 + 9 * j, 8 + 5 * j, 5 + 9 * j + W // 3, 8 + 5 * j + H // 3) for j in range(1 + i % 2))
        (base / "Annotations" / (stem + ".xml")).write_text("<annotation><filename>%s.jpg</filename>%s</annotation>" % (stem, objs))
    (base / "ImageSets" / "Main" / "trainval.txt").write_text("".join(s_ + "\n" for s_ in stems))
    ds = vu.get_voc2007(str(tmp_path), "trainval", None)
    labeled = list(range(len(ds)))
    groups = create_aspect_ratio_groups(ds, k=3)
    assert len(set(groups)) == 2
    torch.manual_seed(3); random.seed(5)
    sampler = GroupedBatchSampler(SubsetRandomSampler(labeled), groups, 2)
    loader = ds.resident_train_loader(sampler, labeled)
    for k, i in enumerate(labeled):                                   # the resident pool is Pillow's decode
        np.testing.assert_array_equal(loader.pool[k].cpu().numpy(), np.asarray(Image.open(ds.images[i]).convert("RGB")))
    seen = []
    for images, targets in loader:
        assert len(images) == 2 and len({im.shape[0] > im.shape[1] for im in images}) == 1      # one orientation per batch
        for im, t in zip(images, targets):
            assert im.is_cuda and im.dtype == torch.uint8 and im.shape[2] == 3
            b = t["boxes"]
            assert bool((b[:, 0] < b[:, 2]).all()) and float(b[:, 2].max()) <= im.shape[1] and float(b[:, 0].min()) >= 0
            seen.append(tuple(im.shape))
    assert len(seen) == 2 * len(loader) == 8
    sd, _, _ = _train_case(torch, n_images=1, seed=2)
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=160, max_size=256).to("cuda")
    model.load_state_dict(sd)
    model.train()
    opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=2e-4, momentum=0.9, weight_decay=1e-4)
    history = engine.train_one_epoch(model, opt, loader, "cuda", cycle=0, epoch=0, print_freq=0)
    assert len(history) == len(loader) == 4 and all(np.isfinite(history))
    assert model._trainer.last_padded_hw in ((160, 224), (160, 256), (224, 160))      # one orientation per batch: never 224 x 224 / 224 x 256
    model.eval()
    assert not np.array_equal(model.state_dict()["roi_heads.box_head.fc7.weight"].numpy(), np.asarray(sd["roi_heads.box_head.fc7.weight"]))


def test_training_batch_with_an_image_without_boxes_and_resnet101(T):
    """Edge cases of the training forward: an image with NO ground-truth boxes (torchvision: every anchor / proposal is background,
    zero regression targets), a batch of one, and the ResNet-101 body: finite losses, gradients for all trainable tensors."""
    torch, ops = T
    from cald_amd import synth, train
    sd, images, targets = _train_case(torch, n_images=2, seed=4)
    targets[1] = {"boxes": torch.zeros((0, 4)), "labels": torch.zeros((0,), dtype=torch.int64)}
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator().manual_seed(2))
    losses = net.forward(images, targets)
    assert all(bool(torch.isfinite(v).all()) for v in losses.values())
    assert int((net.last["roi_labels"][net.last["rois"][:, 0].cpu() == 1] != 0).sum()) == 0      # image 1: background only
    grads = net.backward()
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())
    one = net.forward(images[:1], targets[:1])
    assert all(bool(torch.isfinite(v).all()) for v in one.values())
    sd101 = synth.pseudo_trained_frcnn(21, 101, seed=1)
    net101 = train.FasterRCNNTrainer(sd101, 21, depth=101, min_size=160, max_size=256, generator=torch.Generator().manual_seed(2))
    l101 = net101.forward(images[:1], targets[:1])
    g101 = net101.backward()
    assert len(g101) == 72 + 17 * 3 and all(bool(torch.isfinite(g).all()) for g in g101.values()) and all(bool(torch.isfinite(v).all()) for v in l101.values())
    assert float(g101["backbone.body.layer3.22.conv2.weight"].abs().max()) > 0


def test_focal_and_l1_losses_value_and_gradient(T, oracle):
    """RetinaNet's classification loss (sigmoid focal loss summed over the anchors outside the ignore band, / max(1, #fg) per image,
    mean over images) and its L1 box loss, value + gradient, against the float64 formulas of torchvision.ops.sigmoid_focal_loss."""
    torch, ops = T
    from oracle import torch_train as tt
    g = torch.Generator().manual_seed(8)
    N, A, K, ld = 2, 9, 7, 64
    level_pix = [20, 12, 6, 2, 1]
    A_tot = sum(level_pix) * A
    blocks = [torch.randn(N, n, ld, generator=g) * 2 for n in level_pix]
    flat = torch.cat([b.reshape(-1) for b in blocks])
    matched = torch.randint(-2, 3, (N, A_tot), generator=g, dtype=torch.int32)
    gt_labels = torch.tensor([1, 4, 6, 2, 3, 5], dtype=torch.int64); gt_off = torch.tensor([0, 3, 6], dtype=torch.int32)
    nfg = [(matched[i] >= 0).sum().item() for i in range(N)]
    img_w = torch.tensor([1.0 / (max(1, n) * N) for n in nfg])
    fd = flat.double().requires_grad_()
    want = 0.0
    for i in range(N):
        logits = []
        o = 0
        for n in level_pix:
            blk = fd[o:o + N * n * ld].view(N, n, ld)[i, :, :A * K].reshape(n * A, K); o += N * n * ld
            logits.append(blk)
        logits = torch.cat(logits)
        tgt = torch.zeros_like(logits)
        fg = matched[i] >= 0
        tgt[fg, gt_labels[gt_off[i] + matched[i][fg].long()]] = 1.0
        valid = matched[i] != -2
        want = want + tt.sigmoid_focal_loss_sum(logits[valid], tgt[valid]) / max(1, nfg[i]) / N
    (1.7 * want).backward()
    grad = torch.zeros_like(flat, device="cuda")
    got = ops.focal_loss(flat.cuda(), level_pix, N, A, K, ld, matched.cuda(), gt_labels.cuda(), gt_off.cuda(), img_w.cuda(), grad=grad, gscale=1.7)
    _close(got, want.reshape(1), 2e-6, "focal loss"); _close(grad, fd.grad, 1e-5, "focal loss gradient")
    pred = torch.randn(500, generator=g); idx = torch.arange(0, 480, 8)[:50]; tgt4 = torch.randn(50, 4, generator=g); w = torch.rand(50, generator=g)
    pd = pred.double().requires_grad_()
    sel = torch.stack([pd[idx + j] for j in range(4)], dim=1)
    l1 = ((sel - tgt4.double()).abs().sum(dim=1) * w.double()).sum(); l1.backward()
    gp = torch.zeros(500, device="cuda")
    got = ops.smooth_l1(pred.cuda(), idx.cuda(), tgt4.cuda(), 0.0, 1.0, grad=gp, weights=w.cuda())
    _close(got, l1.reshape(1), 2e-6, "weighted L1"); _close(gp, pd.grad, 1e-6, "weighted L1 gradient")


def test_retinanet_training_step_vs_autograd_and_drop_in(T, oracle):
    """RetinaNet (detection/retinanet_cal.py) training step: both losses and the gradient of every trainable tensor (body layers
    2-4, FPN, P6/P7, both towers, both output convs) against float64 autograd with the reference's own loss code restated
    (oracle/torch_train.py TorchTrainRetinaNet), same ReLU decisions; then the drop-in loop on the HipDetector object."""
    torch, ops = T
    from cald_amd import synth, train, detector
    from oracle import torch_train as tt
    _, images, targets = _train_case(torch, seed=6)
    sd = synth.pseudo_trained_retinanet(21, 50, seed=2)
    net = train.RetinaNetTrainer(sd, 21, min_size=160, max_size=256)
    losses = net.forward(images, targets)
    grads = {k: v.clone() for k, v in net.backward().items()}
    ref = tt.TorchTrainRetinaNet(sd, 21, min_size=160, max_size=256)
    ref.masks = net.relu_decisions()
    want, rec = ref.losses(images, targets)
    assert torch.equal(rec["matched"].int(), torch.from_numpy(net.last["matched_host"])), "same anchor matching"
    for k in want:
        got, w = float(losses[k]), float(want[k].detach())
        assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (k, got, w)
    sum(want.values()).backward()
    tr = ref.trainable()
    assert sorted(tr) == sorted(grads) and len(grads) == 78
    worst = ("", 0.0)
    for k, g in grads.items():
        w = tr[k].grad
        err = float((g.double().cpu() - w).abs().max()) / float(w.abs().max())
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 1e-4, "largest gradient error %.3g at %s" % (worst[1], worst[0])
    # drop-in: the HipDetector in train mode, HIP SGD, then back to inference on the updated weights
    model = detector.retinanet_resnet50_fpn_cal(num_classes=21, min_size=160, max_size=256).to("cuda")
    model.load_state_dict(sd)
    model.train()
    opt = train.SGD([p for p in model.parameters() if p.requires_grad], lr=2e-5, momentum=0.0, weight_decay=1e-4)
    seen = []
    for it in range(6):
        loss_dict = model(images, targets)
        assert sorted(loss_dict) == ["bbox_regression", "classification"]
        losses_sum = sum(loss for loss in loss_dict.values())
        seen.append(float(losses_sum.detach()))
        opt.zero_grad(); losses_sum.backward(); opt.step()
    assert all(np.isfinite(seen)) and seen[-1] < seen[0], seen
    model.eval()
    out = model([images[0]])
    assert set(out[0]) >= {"boxes", "scores", "labels"}


def test_retinanet_loss_kernels_match_the_reference_code(T, golden):
    """The device side of RetinaNet's loss -- `cald_train_match` (0.5 / 0.4, low-quality matches), `cald_train_box_encode`,
    `focal_loss_kernel`, `smooth_l1_kernel` (beta = 0) -- against the REFERENCE's own compute_loss bodies
    (detection/retinanet_cal.py:100-133, :185-223, :389-400; tests/golden/train_losses.npz from
    oracle/make_golden_train_losses.py): matched indices identical, losses within float32 summation noise (2e-6) of the
    reference's float64 run and 1e-5 of its float32 run."""
    torch, ops = T
    g = golden("train_losses")
    for k in range(int(g["l_n"])):
        N, K = int(g["l%d_N" % k]), int(g["l%d_K" % k])
        level_pix = [int(h * w) for h, w in g["l%d_level_hw" % k]]
        anchors = torch.from_numpy(g["l%d_anchors" % k]).cuda()
        cls, reg = g["l%d_cls_logits" % k], g["l%d_bbox_regression" % k]
        ld = (9 * K + 15) // 16 * 16
        cls_blocks, reg_blocks, reg_off, lvl_start = [], [], [], []
        o = ro = 0
        for n in level_pix:
            blk = np.zeros((N, n, ld), np.float32)
            blk[:, :, :9 * K] = cls[:, o:o + 9 * n].reshape(N, n, 9 * K)
            cls_blocks.append(blk.reshape(-1)); reg_blocks.append(reg[:, o:o + 9 * n].reshape(N, n, 36).reshape(-1))
            reg_off.append(ro); lvl_start.append(o)
            ro += N * n * 36; o += 9 * n
        cls_flat = torch.from_numpy(np.concatenate(cls_blocks)).cuda(); reg_flat = torch.from_numpy(np.concatenate(reg_blocks)).cuda()
        matched, gts, labels = [], [], []
        for i in range(N):
            gt = torch.from_numpy(g["l%d_gt%d" % (k, i)]).cuda()
            gts.append(gt); labels.append(torch.from_numpy(g["l%d_labels%d" % (k, i)]))
            matched.append(ops.match(anchors, gt, 0.5, 0.4, True))
        m = torch.stack(matched)
        assert np.array_equal(m.cpu().numpy().astype(np.int64), g["l%d_matched" % k]), "case %d: matcher decisions" % k
        mh = m.cpu().numpy()
        gt_off = np.concatenate([[0], np.cumsum([b.shape[0] for b in gts])]).astype(np.int32)
        nfg = [(mh[i] >= 0).sum() for i in range(N)]
        img_w = torch.tensor([1.0 / (max(1, n) * N) for n in nfg], dtype=torch.float32).cuda()
        got_c = ops.focal_loss(cls_flat, level_pix, N, 9, K, ld, m.contiguous(), torch.cat(labels).cuda(), torch.from_numpy(gt_off).cuda(), img_w)
        box_idx, anc, gsel, wts = [], [], [], []
        ls = np.array(lvl_start)
        for i in range(N):
            fg = np.nonzero(mh[i] >= 0)[0]
            l = np.searchsorted(ls, fg, side="right") - 1
            rel = fg - ls[l]
            pix, a = rel // 9, rel % 9
            box_idx.append(np.array(reg_off)[l] + (i * np.array(level_pix)[l] + pix) * 36 + 4 * a)
            anc.append(fg); gsel.append(gt_off[i] + mh[i][fg]); wts.append(np.full(len(fg), 1.0 / (max(1, len(fg)) * N), np.float32))
        box_idx, anc, gsel = [torch.from_numpy(np.concatenate(v).astype(np.int64)).cuda() for v in (box_idx, anc, gsel)]
        tgt = ops.box_encode(torch.cat(gts)[gsel].contiguous(), anchors[anc].contiguous(), (1.0, 1.0, 1.0, 1.0))
        got_r = ops.smooth_l1(reg_flat, box_idx, tgt, 0.0, 1.0, weights=torch.from_numpy(np.concatenate(wts)).cuda())
        for got, name in ((got_c, "cls"), (got_r, "reg")):
            w64, w32 = float(g["l%d_%s_f64" % (k, name)]), float(g["l%d_%s_f32" % (k, name)])
            assert abs(float(got) - w64) <= 2e-6 * abs(w64), (k, name, float(got), w64)
            assert abs(float(got) - w32) <= 1e-5 * abs(w32), (k, name, float(got), w32)


def test_randperm_sampler_reproduces_torchvision_draws(T, oracle):
    """sampler="randperm": the trainer consumes ``torch.randperm(n, generator=g)[:k]`` exactly as torchvision's
    BalancedPositiveNegativeSampler does (positives then negatives, image by image, RPN before the RoI heads), so the float64
    checker -- which draws ITS OWN samples from an equally seeded generator (oracle/torch_train.sample) -- lands on the same
    anchors and the same RoIs without anything being handed over, and on the same losses."""
    torch, ops = T
    from cald_amd import train
    from oracle import torch_train as tt
    sd, images, targets = _train_case(torch, n_images=3, seed=4)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(11), sampler="randperm")
    losses = net.forward(images, targets)
    props = [p.cpu() for p in net.last["proposals"]]
    ref = tt.TorchTrainFRCNN(sd, 21, min_size=160, max_size=256)
    ref.masks = net.relu_decisions()
    want, rec = ref.losses(images, targets, props, torch.Generator().manual_seed(11), cfg=dict(box_batch=64), samples=None)
    A = rec["anchors"].shape[0]
    got_pos = np.concatenate([i * A + sp for i, (sp, sn) in enumerate(net.last["samples"]["rpn"])])
    got_neg = np.concatenate([i * A + sn for i, (sp, sn) in enumerate(net.last["samples"]["rpn"])])
    assert np.array_equal(got_pos, rec["rpn_pos"].numpy()) and np.array_equal(got_neg, rec["rpn_neg"].numpy()), "RPN samples"
    assert torch.equal(rec["roi_labels"], net.last["roi_labels"]), "sampled RoIs"
    for k in want:
        got, w = float(losses[k]), float(want[k].detach())
        assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (k, got, w)
    # the default sampler draws differently from the same seed (it never permutes all candidates)
    other = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(11))
    other.forward(images, targets)
    assert not all(np.array_equal(a[1], b[1]) for a, b in zip(other.last["samples"]["rpn"], net.last["samples"]["rpn"]))
    with pytest.raises(ValueError):
        train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, sampler="nope")


def test_full_size_training_step_gradients_vs_autograd(T, oracle):
    """cald_train.py's training defaults AT SIZE: min_size 600 / max_size 1000, 2 000 proposals, 512 sampled RoIs per image (batch 2 here --
    the float64 checker of a batch of 4 alone took 160 s of the suite's 20 minutes; the reference's batch 4 is what bench.py's training leg
    and the small-size tests run).  The four losses and the gradient of every one of the 72 trainable tensors against float64 torch-CPU
    autograd (oracle/torch_train.py), randperm sampler so the checker draws its own samples.  Tolerance 1e-4 (float32 vs float64)."""
    import time
    torch, ops = T
    from cald_amd import train
    from oracle import torch_train as tt
    t0 = time.time()
    sd, images, targets = _train_case(torch, n_images=2, seed=9, scale=1.0)
    assert max(max(im.shape[1:]) for im in images) == 500
    net = train.FasterRCNNTrainer(sd, 21, min_size=600, max_size=1000, generator=torch.Generator().manual_seed(21), sampler="randperm")
    losses = net.forward(images, targets)
    props = [p.cpu() for p in net.last["proposals"]]
    assert all(p.shape[0] > 1000 for p in props)
    grads = {k: v.clone() for k, v in net.backward().items()}
    assert net.last["roi_labels"].numel() == 2 * 512
    ref = tt.TorchTrainFRCNN(sd, 21, min_size=600, max_size=1000)
    ref.masks = net.relu_decisions()
    want, rec = ref.losses(images, targets, props, torch.Generator().manual_seed(21))
    assert torch.equal(rec["roi_labels"], net.last["roi_labels"]), "same sampled RoIs"
    for k in want:
        got, w = float(losses[k]), float(want[k].detach())
        assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (k, got, w)
    sum(want.values()).backward()
    tr = ref.trainable()
    assert sorted(tr) == sorted(grads) and len(grads) == 72
    worst = ("", 0.0)
    for k, g in grads.items():
        w = tr[k].grad
        err = float((g.double().cpu() - w).abs().max()) / float(w.abs().max())
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 1e-4, "largest gradient error %.3g at %s" % (worst[1], worst[0])
    print("full-size training parity: worst gradient error %.3g at %s, %.0f s" % (worst[1], worst[0], time.time() - t0))


def test_sgd_resume_uses_the_loaded_momentum(T):
    """optimizer.load_state_dict() after fused steps (resume): the loaded momentum buffers are what the next fused step uses,
    i.e. the trajectory equals torch.optim.SGD resumed from the same state."""
    torch, ops = T
    from cald_amd import train
    sd, images, targets = _train_case(torch)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(7))
    params = list(net.parameters())
    opt = train.SGD(params, lr=1e-3, momentum=0.9, weight_decay=1e-4, net=net)
    mirror = [torch.nn.Parameter(p.detach().clone()) for p in params]
    ref = torch.optim.SGD(mirror, lr=1e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(0)

    def step_both(seed_scale):
        for p, m in zip(params, mirror):
            gr = torch.randn(p.shape, generator=g).to(p.device) * seed_scale
            if p.grad is None:
                p.grad = net.grads[[k for k in net.names if net.params[k] is p][0]]
            p.grad.copy_(gr); m.grad = gr.clone()
        opt.step(); ref.step()
    step_both(1.0); step_both(0.5)
    assert opt._mflat is not None                       # the fused path ran
    state = ref.state_dict()
    # "resume": momentum from a checkpoint that differs from what the optimizer currently holds
    for st in state["state"].values():
        st["momentum_buffer"] = st["momentum_buffer"] * 3.0 + 0.25
    ref.load_state_dict(state); opt.load_state_dict({"state": {i: {"momentum_buffer": v["momentum_buffer"].clone()} for i, v in state["state"].items()},
                                                      "param_groups": opt.state_dict()["param_groups"]})
    step_both(0.25)
    for p, m in zip(params, mirror):
        _close(p, m, 1e-6, "parameter after the resumed step")
    assert all(opt.state[p]["momentum_buffer"].data_ptr() == opt._mflat.data_ptr() + 4 * net._off[k] for p, k in zip(params, net.names))


def test_full_size_retinanet_training_step_gradients_vs_autograd(T, oracle):
    """RetinaNet at the reference's training size (min 600 / max 1000, cald_train.py:342), two images: both losses, the anchor
    matching and the gradient of every one of the 78 trainable tensors against float64 torch-CPU autograd whose loss code is pinned to
    the reference's own (tests/golden/train_losses.npz).  Tolerance 1e-4 (float32 vs float64)."""
    torch, ops = T
    from cald_amd import synth, train
    from oracle import torch_train as tt
    _, images, targets = _train_case(torch, n_images=2, seed=12, scale=1.0)
    sd = synth.pseudo_trained_retinanet(21, 50, seed=2)
    net = train.RetinaNetTrainer(sd, 21, min_size=600, max_size=1000)
    losses = net.forward(images, targets)
    grads = {k: v.clone() for k, v in net.backward().items()}
    ref = tt.TorchTrainRetinaNet(sd, 21, min_size=600, max_size=1000)
    ref.masks = net.relu_decisions()
    want, rec = ref.losses(images, targets)
    assert torch.equal(rec["matched"].int(), torch.from_numpy(net.last["matched_host"])), "same anchor matching"
    assert int((rec["matched"] >= 0).sum()) > 50
    for k in want:
        got, w = float(losses[k]), float(want[k].detach())
        assert abs(got - w) <= 1e-4 * max(1.0, abs(w)), (k, got, w)
    sum(want.values()).backward()
    tr = ref.trainable()
    assert sorted(tr) == sorted(grads) and len(grads) == 78
    worst = ("", 0.0)
    for k, g in grads.items():
        w = tr[k].grad
        err = float((g.double().cpu() - w).abs().max()) / float(w.abs().max())
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 1e-4, "largest gradient error %.3g at %s" % (worst[1], worst[0])


def test_training_input_and_ordering_errors_are_loud(T):
    """torchvision's GeneralizedRCNN.forward rejects degenerate target boxes with a ValueError naming the box; the layers keep ONE set of
    saved activations, so backward() of a forward that a later forward has replaced must refuse instead of differentiating the wrong
    tensors; load_state_dict() in train mode leaves a working trainer behind."""
    torch, ops = T
    from cald_amd import detector, train
    sd, images, targets = _train_case(torch)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, box_batch=64, generator=torch.Generator().manual_seed(7))
    bad = [dict(t) for t in targets]
    bad[1] = {"boxes": bad[1]["boxes"].clone(), "labels": bad[1]["labels"]}
    bad[1]["boxes"][0, 2] = bad[1]["boxes"][0, 0]                      # zero width
    with pytest.raises(ValueError, match="positive height and width"):
        net.forward(images, bad)
    model = train.TrainableDetector(net)
    first = sum(model(images, targets).values())
    second = sum(model(images, targets).values())
    with pytest.raises(RuntimeError, match="later forward"):
        first.backward()
    second.backward()                                                   # the latest forward differentiates normally
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)
    det = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=160, max_size=256).to("cuda")
    det.load_state_dict(sd)
    det.train()
    det.load_state_dict(sd)                                             # in train mode: the trainer is rebuilt, the next call works
    out = det(images, targets)
    assert sorted(out) == ["loss_box_reg", "loss_classifier", "loss_objectness", "loss_rpn_box_reg"]


_SEG_CAP_SCRIPT = r"""
import sys, json
import numpy as np, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from cald_amd import train, _ffi
import test_gpu_train as tg
out = []
for scale, seed in ((0.4, 21), (0.3, 22), (0.4, 21)):      # three geometries' worth of tables, then the first one again
    sd, images, targets = tg._train_case(torch, n_images=2, seed=seed, scale=scale)
    net = train.FasterRCNNTrainer(sd, 21, min_size=160, max_size=256, generator=torch.Generator().manual_seed(5))
    losses = net.forward(images, targets)
    grads = net.backward()
    torch.cuda.synchronize()
    out.append([float(v) for v in losses.values()] + [float(g.double().abs().sum()) for g in grads.values()])
print("RESULT " + json.dumps({"runs": out, "tables": _ffi.lib().cald_train_seg_cache_size()}))
"""


def test_geometry_table_cache_eviction_never_frees_inside_an_operator(T):
    """the bounded cache of dense-batch geometry tables used to free every table from INSIDE dense_seg(), while the
    operator that called it still held pointers from its earlier dense_seg() calls.  Eviction now happens only at the next operator's
    entry.  With the bound lowered to 2 tables (every operator with more than two geometries overflows it) three training steps give
    exactly the losses and gradient sums of the default bound."""
    import json, os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
    script = _SEG_CAP_SCRIPT % (root, here)

    def run(cap):
        env = dict(os.environ)
        env.pop("CALD_SEG_CACHE_CAP", None)
        if cap:
            env["CALD_SEG_CACHE_CAP"] = str(cap)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
        return json.loads(line[len("RESULT "):])
    base, small = run(0), run(2)
    assert small["runs"] == base["runs"]
    assert small["runs"][0] == small["runs"][2]          # same inputs before and after the evictions in between
    assert small["tables"] < base["tables"]              # the cache was emptied on the way (the bound holds up to the tables of one operator call)


def test_frcnn_loss_kernels_match_the_reference_repo_copies(T, golden):
    """`softmax_ce_kernel`, `smooth_l1_kernel` and `bce_logits_kernel` against the copies of torchvision's Faster R-CNN losses that the
    reference tree itself holds and that were executed from it (detection/frcnn_ll.py:28-63 `_fastrcnn_loss`, :245-281
    `RegionProposalNetwork._compute_loss`; tests/golden/frcnn_losses.npz from oracle/make_golden_frcnn_losses.py): the gather of the
    class-specific deltas of the positive rows, the normalisers (all rows / all sampled anchors), the objectness labels.  Single-image
    cases, where the copies' per-image losses ARE the batch losses; run at the copies' beta (1 for the box head, L1 for the RPN -- the
    hot path's torchvision uses 1/9 for both, the kernels take beta as an argument).  2e-6 of the float64 run, 1e-5 of the float32 run."""
    torch, ops = T
    g = golden("frcnn_losses")

    def close(got, key):
        w64, w32 = float(g[key + "_f64"]), float(g[key + "_f32"])
        assert abs(float(got) - w64) <= 2e-6 * max(abs(w64), 1e-3), (key, float(got), w64)
        assert abs(float(got) - w32) <= 1e-5 * max(abs(w32), 1e-3), (key, float(got), w32)
    for k in range(int(g["b_n"])):
        logits, deltas, labels, tgt = g["b%d_logits" % k], g["b%d_deltas" % k], g["b%d_labels" % k], g["b%d_targets" % k]
        R, Cc = logits.shape
        pred = torch.from_numpy(np.concatenate([logits, deltas], axis=1)).cuda().contiguous()          # the fused predictor's rows: C logits, 4 C deltas
        ld = pred.shape[1]
        close(ops.softmax_ce(pred, torch.from_numpy(labels).cuda(), Cc), "b%d_cls" % k)
        pos = np.flatnonzero(labels > 0)
        idx = torch.from_numpy((pos * ld + Cc + 4 * labels[pos]).astype(np.int64)).cuda()
        got = ops.smooth_l1(pred, idx, torch.from_numpy(tgt[pos]).cuda().contiguous(), 1.0, R) if len(pos) else torch.zeros(1)
        close(got, "b%d_box" % k)
    for k in range(int(g["r_n"])):
        obj, deltas, tgt, pos, neg = g["r%d_obj" % k], g["r%d_deltas" % k], g["r%d_targets" % k], g["r%d_pos" % k], g["r%d_neg" % k]
        A = obj.shape[0]
        head = torch.from_numpy(np.concatenate([obj, deltas], axis=1)).cuda().contiguous()               # [A][1 logit + 4 deltas]
        samp = np.concatenate([pos, neg])
        lab = torch.from_numpy(np.concatenate([np.ones(len(pos), np.float32), np.zeros(len(neg), np.float32)])).cuda()
        close(ops.bce_logits(head, torch.from_numpy((samp * 5).astype(np.int64)).cuda(), lab), "r%d_obj" % k)
        got = (ops.smooth_l1(head, torch.from_numpy((pos * 5 + 1).astype(np.int64)).cuda(), torch.from_numpy(tgt[pos]).cuda().contiguous(), 0.0, len(samp))
               if len(pos) else torch.zeros(1))
        close(got, "r%d_box" % k)
