"""GPU parity of the backbone kernels (conv / depthwise / BN+act / LayerNorm / pooling / SE / residual /
optimiser / augmentation) through the C ABI against torch-CPU float64 references of the same ops.
Tolerance: 2e-5 relative to the tensor's max magnitude (fp32 accumulation-order differences)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from semiuhpe_amd import kernels
    return kernels


def rel(x, ref):
    x = x.detach().cpu().double()
    ref = ref.detach().double()
    return float((x - ref).abs().max() / max(1.0, float(ref.abs().max())))


def nhwc(t):  # (N,C,H,W) cpu -> (N,H,W,C) cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def from_nhwc(t):
    return t.cpu().permute(0, 3, 1, 2)


CONVS = [
    # n, cin, cout, k, stride, pad, groups, h, w
    (2, 3, 24, 3, 2, 1, 1, 32, 32),        # RGB stem (scalar loaders)
    (3, 24, 24, 3, 1, 1, 1, 16, 16),
    (2, 24, 96, 3, 2, 1, 1, 16, 16),
    (2, 96, 48, 1, 1, 0, 1, 9, 9),
    (2, 64, 256, 3, 1, 1, 1, 14, 14),
    (2, 128, 128, 3, 1, 1, 2, 14, 14),     # RepVGG g2
    (2, 128, 128, 1, 1, 0, 2, 14, 14),
    (2, 64, 128, 3, 2, 1, 1, 15, 15),      # odd extent, stride 2
    (2, 64, 128, 1, 2, 0, 1, 14, 14),      # RepVGG 1x1 stride-2 branch
    (5, 256, 1280, 1, 1, 0, 1, 7, 7),
    (2, 64, 64, 3, 1, 1, 64, 14, 14),      # depthwise
    (2, 64, 64, 3, 2, 1, 64, 14, 14),
    (2, 512, 512, 3, 2, 1, 512, 15, 15),
    (3, 1536, 1536, 3, 1, 1, 1536, 7, 7),  # > 256 channel vectors: two column chunks
    (40, 160, 960, 1, 1, 0, 1, 14, 14),    # larger M: wide tiles
    (33, 48, 192, 3, 1, 1, 1, 28, 28),
    (40, 960, 160, 1, 1, 0, 1, 14, 14),    # 160-wide tiles (fwd columns, wgrad rows)
    (20, 24, 96, 3, 2, 1, 1, 56, 56),      # 96-wide tile
    (20, 96, 48, 1, 1, 0, 1, 28, 28),
]


@pytest.mark.parametrize("cfg", CONVS, ids=[str(c) for c in CONVS])
def test_conv_fwd_dgrad_wgrad(K, cfg):
    n, cin, cout, k, stride, pad, groups, h, w = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, None, stride, pad, 1, groups)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    wg = K.to_filter_layout(wt.cuda())
    xg = nhwc(x)
    y = K.conv2d_fwd(xg, wg, None, stride, pad, groups)
    assert rel(from_nhwc(y), y_ref) < 2e-5
    dyg = nhwc(dy)
    dx = K.conv2d_dgrad(dyg, wg, xg.shape, stride, pad, groups)
    assert rel(from_nhwc(dx), xd.grad) < 2e-5
    dw = K.conv2d_wgrad(xg, dyg, wg.shape, stride, pad, groups)
    assert rel(dw, wd.grad) < 2e-5
    dw2 = K.conv2d_wgrad(xg, dyg, wg.shape, stride, pad, groups, out=dw.clone(), accumulate=True)
    assert rel(dw2, 2 * wd.grad) < 2e-5


@pytest.mark.parametrize("cfg", [(6, 48, 192, 3, 1, 1, 1, 28, 28), (40, 160, 960, 1, 1, 0, 1, 14, 14), (4, 64, 128, 3, 2, 1, 1, 30, 30),
                                 (2, 128, 128, 3, 1, 1, 2, 14, 14)], ids=str)
def test_tile_menu_every_pinned_tile_matches(K, cfg):
    """suhpe_conv2d_set_tile: every tile of the menu gives the same forward / input gradient bit for bit (the k order of an
    output element does not depend on the tile) and the same filter gradient within fp32 summation-order noise; the
    autotuner (kernels._autotune) then pins one of them."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    n, cin, cout, k, stride, pad, groups, h, w = cfg
    g = torch.Generator().manual_seed(11)
    x = nhwc(torch.randn(n, cin, h, w, generator=g))
    wg = K.to_filter_layout((torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)).cuda())
    d = K.conv_desc(x.shape, wg.shape, stride, pad, groups)
    y0 = K.conv2d_fwd(x, wg, None, stride, pad, groups)
    dy = torch.randn_like(y0)
    dx0 = K.conv2d_dgrad(dy, wg, x.shape, stride, pad, groups)
    dw0 = K.conv2d_wgrad(x, dy, wg.shape, stride, pad, groups)
    try:
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
            for tile in range(lib.suhpe_conv2d_tile_count(pid)):
                wino = b"wino" in lib.suhpe_conv2d_tile_name(pid, tile)
                halo = b"nt3x3" in lib.suhpe_conv2d_tile_name(pid, tile) or (wino and pid < 2)      # (Winograd: other arithmetic, rounding-level differences)
                resident = b"resident" in lib.suhpe_conv2d_tile_name(pid, tile)
                dma = b"gemm_nt_dma" in lib.suhpe_conv2d_tile_name(pid, tile)
                if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) != 0:
                    # the scalar-loader tile is refused for vector shapes, the halo / Winograd tiles for anything but 3x3 / stride 1 / pad 1,
                    # the resident and LDS-DMA tiles for anything but 1x1 / stride 1 (the DMA tiles also for reductions that are not multiples of 32,
                    # the narrow one for more than 64 outputs)
                    assert (pid < 2 and (tile == 0 or halo or resident or dma)) or wino
                    continue
                assert not wino or (k == 3 and stride == 1)
                assert not (resident or dma) or k == 1
                if halo:            # channel-chunk-major reduction order: same sums, different rounding
                    out = K.conv2d_fwd(x, wg, None, stride, pad, groups) if pid == 0 else K.conv2d_dgrad(dy, wg, x.shape, stride, pad, groups)
                    assert rel(out, (y0 if pid == 0 else dx0).cpu()) < 3e-6, tile
                elif pid == 0:
                    assert torch.equal(K.conv2d_fwd(x, wg, None, stride, pad, groups), y0), tile
                elif pid == 1:
                    assert torch.equal(K.conv2d_dgrad(dy, wg, x.shape, stride, pad, groups), dx0), tile
                else:
                    assert rel(K.conv2d_wgrad(x, dy, wg.shape, stride, pad, groups), dw0.cpu()) < 2e-5, tile
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, 99) != 0
        # autotuner: pins something valid and leaves results unchanged
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
        K._tuned.clear()
        K.AUTOTUNE = True
        assert rel(K.conv2d_fwd(x, wg, None, stride, pad, groups), y0.cpu()) < 3e-6         # (bit-equal unless a halo tile wins)
        assert rel(K.conv2d_dgrad(dy, wg, x.shape, stride, pad, groups), dx0.cpu()) < 3e-6
        assert rel(K.conv2d_wgrad(x, dy, wg.shape, stride, pad, groups), dw0.cpu()) < 2e-5
        assert len(K._tuned) == 3
    finally:
        K.AUTOTUNE = False
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)


@pytest.mark.parametrize("n,c,hw,stride,act,seg", [(8, 96, 14, 1, "silu", None), (40, 64, 14, 1, "silu", 8), (3, 32, 15, 2, "silu", 1), (2, 24, 9, 1, "gelu", None),
                                                   (64, 128, 7, 1, "silu", 13), (160, 256, 14, 1, "silu", 32), (6, 20, 6, 1, "silu", 4),
                                                   (6, 96, 28, 1, "gelu", 2), (3, 64, 56, 1, "gelu", None), (4, 32, 28, 1, "silu", 1)])
def test_depthwise_with_folded_batchnorm_input(K, n, c, hw, stride, act, seg):
    """suhpe_dwconv_bnact_fwd / _wgrad: depthwise conv (and its filter gradient) of act(BN(x)) computed from x while loading
    == the unfused bn_act_fwd -> depthwise chain (zero padding applied after the activation); with batch segments each
    segment is normalised with its own statistics; the statistics epilogue of the output rides along unchanged"""
    g = torch.Generator().manual_seed(n + c)
    x = (torch.randn(n, hw, hw, c, generator=g) * 1.3 + 0.2).cuda()
    if seg:
        x[seg:] = x[seg:] * 0.6 - 0.4                       # the two segments have visibly different statistics
        mean, rstd = K.bn_stats_seg(x, seg * hw * hw, c, 1e-3, 0.1)
    else:
        mean, rstd = K.bn_stats(x, c, 1e-3, 0.1)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    w = K.to_filter_layout((torch.randn(c, 1, 3, 3, generator=g) / 3).cuda())
    if seg:
        y_act = K.bn_act_fwd_seg(x, seg * hw * hw, c, mean, rstd, gamma, beta, act)
    else:
        y_act = K.bn_act_fwd(x, c, mean, rstd, gamma, beta, act)
    ref = K.conv2d_fwd(y_act, w, None, stride, 1, c)
    out = K.dwconv_bnact_fwd(x, mean, rstd, gamma, beta, act, w, stride, 1, seg=seg)
    assert rel(out, ref.cpu()) < 4e-6                       # SiLU on v_exp / v_rcp: ~1 ulp each
    dy = torch.randn(ref.shape, generator=g).cuda()
    dw_ref = K.conv2d_wgrad(y_act, dy, w.shape, stride, 1, c)
    dw = K.dwconv_bnact_wgrad(x, mean, rstd, gamma, beta, act, dy, w.shape, stride, 1, seg=seg)
    assert rel(dw, dw_ref.cpu()) < 2e-5
    if K.dwconv_stats_blocks(x.shape, w.shape, stride, 1):
        res = K.dwconv_bnact_fwd(x, mean, rstd, gamma, beta, act, w, stride, 1, seg=seg, stats=True)
        assert rel(res[0], out.cpu()) < 1e-6                # the STATS instantiation contracts one multiply-add differently
        rows = out.numel() // c
        if seg:
            rows_a = seg * out.shape[1] * out.shape[2]
            m2, r2 = K.bn_stats_from_partials_seg(res[1], res[2], res[3][0], res[3][1], rows, rows_a, c, 1e-3, 0.1)
            m0, r0 = K.bn_stats_seg(out, rows_a, c, 1e-3, 0.1)
        else:
            m2, r2 = K.bn_stats_from_partials(res[1], res[2], rows, c, 1e-3, 0.1)
            m0, r0 = K.bn_stats(out, c, 1e-3, 0.1)
        assert (m2 - m0).abs().max().item() < 2e-6 and rel(r2, r0.cpu()) < 2e-6


def test_deferred_wgrad_reduction_is_bit_identical(K):
    """kernels.deferred_wgrads: one batched slab reduction at the end of a backward pass == the layer-by-layer reductions"""
    g = torch.Generator().manual_seed(3)
    cases = []
    for (n, cin, cout, k, stride, pad, groups, h, w) in [(6, 48, 192, 3, 1, 1, 1, 28, 28), (40, 160, 960, 1, 1, 0, 1, 14, 14), (4, 64, 128, 3, 2, 1, 1, 30, 30),
                                                         (2, 128, 128, 3, 1, 1, 2, 14, 14), (16, 512, 128, 1, 1, 0, 1, 1, 1)]:
        x = nhwc(torch.randn(n, cin, h, w, generator=g))
        wshape = (cout, cin // groups, k, k)
        ho = (h + 2 * pad - k) // stride + 1
        dy = torch.randn(n, ho, ho, cout, generator=g).cuda()
        g0 = K.to_filter_layout(torch.randn(wshape, generator=g).cuda())
        cases.append((x, dy, wshape, stride, pad, groups, g0))
    ref = []
    for x, dy, wshape, stride, pad, groups, g0 in cases:
        ref.append(K.conv2d_wgrad(x, dy, wshape, stride, pad, groups, out=g0.clone(), accumulate=True))
    outs = [c[6].clone() for c in cases]
    with K.deferred_wgrads():
        for (x, dy, wshape, stride, pad, groups, _), o in zip(cases, outs):
            K.conv2d_wgrad(x, dy, wshape, stride, pad, groups, out=o, accumulate=True)
        assert not torch.equal(outs[0], ref[0])            # nothing reduced yet
    for o, r in zip(outs, ref):
        assert torch.equal(o, r)


STAT_CONVS = [c for c in CONVS if not (c[6] > 1 and c[6] == c[1] == c[2])] + [
    (1, 24, 24, 3, 1, 1, 1, 5, 5),         # a single ragged row tile
    (160, 24, 24, 3, 1, 1, 1, 28, 28),     # many row tiles, narrow output (256x32 / 128x32 tiles)
    (37, 256, 64, 1, 1, 0, 1, 13, 13),     # M not a multiple of any tile height
    (16, 64, 128, 3, 2, 1, 2, 28, 28),     # grouped + strided
]


@pytest.mark.parametrize("cfg", STAT_CONVS, ids=[str(c) for c in STAT_CONVS])
def test_conv_fwd_fused_bn_statistics(K, cfg):
    """conv2d_fwd_stats == conv2d_fwd bit for bit, and its epilogue partials finish to the same batch statistics /
    running-stat update as the stand-alone bn_stats pass (and as a float64 reference)."""
    n, cin, cout, k, stride, pad, groups, h, w = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000 + 1)
    x = torch.randn(n, cin, h, w, generator=g) + 0.3
    wt = torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)
    wg, xg = K.to_filter_layout(wt.cuda()), nhwc(x)
    y0 = K.conv2d_fwd(xg, wg, None, stride, pad, groups)
    y1, part, tiles = K.conv2d_fwd_stats(xg, wg, stride, pad, groups)
    assert torch.equal(y0, y1)
    assert part.numel() == cout * tiles * 4
    eps, mom = 1e-3, 0.1
    rows = y0.numel() // cout

    def fresh():
        return (torch.full((cout,), 0.25, device="cuda"), torch.full((cout,), 1.5, device="cuda"),
                torch.zeros((), dtype=torch.int64, device="cuda"))
    rm0, rv0, nb0 = fresh()
    rm1, rv1, nb1 = fresh()
    m0, r0 = K.bn_stats(y0, cout, eps, mom, rm0, rv0, nb0)
    m1, r1 = K.bn_stats_from_partials(part, tiles, rows, cout, eps, mom, rm1, rv1, nb1)
    yd = y0.double().cpu().reshape(rows, cout)
    mean = yd.mean(0)
    var = yd.var(0, unbiased=False)
    assert rel(m1, mean) < 2e-5 and rel(r1, (var + eps).rsqrt()) < 2e-5
    assert rel(m1, m0.cpu()) < 2e-6 and rel(r1, r0.cpu()) < 2e-5
    assert rel(rm1, rm0.cpu()) < 2e-6 and rel(rv1, rv0.cpu()) < 2e-5
    assert rel(rv1, 0.9 * 1.5 + 0.1 * yd.var(0, unbiased=rows > 1)) < 2e-5
    assert int(nb1) == int(nb0) == 1


@pytest.mark.parametrize("cfg", [(4, 24, 24, 3, 1, 1, 1, 28, 28), (3, 96, 48, 1, 1, 0, 1, 14, 14), (2, 128, 128, 3, 1, 1, 2, 14, 14),
                                 (2, 64, 128, 3, 2, 1, 1, 15, 15), (40, 160, 960, 1, 1, 0, 1, 14, 14)], ids=str)
@pytest.mark.parametrize("act,res", [("silu", 0), ("none", 2), ("relu", 1), ("silu", 2)])
def test_conv_fwd_folded_eval_batchnorm(K, cfg, act, res):
    """conv -> eval BatchNorm -> activation (-> skip add before / after the activation) folded into the GEMM epilogue"""
    n, cin, cout, k, stride, pad, groups, h, w = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000 + 7)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)
    rm, rv = torch.randn(cout, generator=g) * 0.3, torch.rand(cout, generator=g) + 0.4
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    eps = 1e-3
    t = F.conv2d(x.double(), wt.double(), None, stride, pad, 1, groups)
    r = torch.randn(t.shape, generator=g)
    u = F.batch_norm(t, rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.0, eps)
    fn = {"silu": F.silu, "relu": F.relu, "none": lambda z: z}[act]
    ref = fn(u + r.double()) if res == 1 else (fn(u) + r.double() if res == 2 else fn(u))
    scale, shift = K.bn_eval_affine(rm.cuda(), rv.cuda(), gamma.cuda(), beta.cuda(), eps)
    assert rel(scale, gamma.double() / (rv.double() + eps).sqrt()) < 1e-6
    y = K.conv2d_fwd_affine(nhwc(x), K.to_filter_layout(wt.cuda()), scale, shift, act, nhwc(r) if res else None, res == 2, stride, pad, groups)
    assert rel(from_nhwc(y), ref) < 2e-5


@pytest.mark.parametrize("c,groups,identity,stride", [(64, 1, True, 1), (128, 2, True, 1), (64, 2, False, 2), (48, 1, False, 1)])
def test_repvgg_reparameterised_block(K, c, groups, identity, stride):
    """suhpe_repvgg_fuse + one conv(+bias, ReLU) == relu(bn3(conv3 x) + bn1(conv1 x) + bn_id(x)) in eval mode (repvgg.py:67-77, 104-142)"""
    g = torch.Generator().manual_seed(c + groups)
    cout = c if identity else c * 2
    x = torch.randn(3, c, 14, 14, generator=g)
    w3 = torch.randn(cout, c // groups, 3, 3, generator=g) / (3 * (c // groups) ** 0.5)
    w1 = torch.randn(cout, c // groups, 1, 1, generator=g) / ((c // groups) ** 0.5)

    def bn(ch):
        return [torch.rand(ch, generator=g) + 0.5, torch.randn(ch, generator=g) * 0.2, torch.randn(ch, generator=g) * 0.3, torch.rand(ch, generator=g) + 0.4]
    b3, b1, bi = bn(cout), bn(cout), (bn(c) if identity else None)
    eps = 1e-5
    xd = x.double()
    f = lambda t, b: F.batch_norm(t, b[2].double(), b[3].double(), b[0].double(), b[1].double(), False, 0.0, eps)
    ref = f(F.conv2d(xd, w3.double(), None, stride, 1, 1, groups), b3) + f(F.conv2d(xd, w1.double(), None, stride, 0, 1, groups), b1)
    if identity:
        ref = ref + f(xd, bi)
    ref = F.relu(ref)
    cu = lambda b: [t.cuda() for t in b] if b is not None else None
    w_eq, ones, b_eq = K.repvgg_fuse(K.to_filter_layout(w3.cuda()), K.to_filter_layout(w1.cuda()), cu(b3), cu(b1), cu(bi), eps)
    y = K.conv2d_fwd_affine(nhwc(x), w_eq, ones, b_eq, "relu", None, False, stride, 1, groups)
    assert rel(from_nhwc(y), ref) < 2e-5


@pytest.mark.parametrize("n,cin,cout", [(160, 1280, 512), (32, 512, 128), (160, 128, 9), (7, 576, 128), (160, 40, 960)])
def test_linear(K, n, cin, cout):
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, cin, generator=g)
    wt = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    xd, wd, bd = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = F.linear(xd, wd, bd)
    dy = torch.randn(n, cout, generator=g)
    y_ref.backward(dy.double())
    xg = x.cuda().view(n, 1, 1, cin)
    wg = wt.cuda().view(cout, cin, 1, 1)
    y = K.conv2d_fwd(xg, wg, b.cuda(), 1, 0, 1)
    assert rel(y.view(n, cout), y_ref) < 2e-5
    dyg = dy.cuda().view(n, 1, 1, cout)
    assert rel(K.conv2d_dgrad(dyg, wg, xg.shape, 1, 0, 1).view(n, cin), xd.grad) < 2e-5
    assert rel(K.conv2d_wgrad(xg, dyg, wg.shape, 1, 0, 1).view(cout, cin), wd.grad) < 2e-5
    assert rel(K.bias_grad(dyg, cout), bd.grad) < 2e-5


def act_ref(u, act):
    return {"none": lambda v: v, "relu": F.relu, "relu6": F.relu6, "silu": F.silu, "gelu": F.gelu,
            "sigmoid": torch.sigmoid}[act](u)


@pytest.mark.parametrize("shape,act", [((6, 14, 14, 24), "silu"), ((4, 7, 7, 1280), "silu"), ((64, 512), "relu6"),
                                         ((3, 9, 9, 48), "none"), ((5, 6, 6, 64), "relu"), ((2, 5, 5, 6), "gelu"),
                                         ((160, 56, 56, 24), "silu")])
def test_batchnorm_act_train(K, shape, act):
    c = shape[-1]
    g = torch.Generator().manual_seed(c)
    x = torch.randn(*shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    dy = torch.randn(*shape, generator=g)
    xd = x.double().reshape(-1, c).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    eps, mom = 1e-3, 0.1
    u = F.batch_norm(xd, rmd, rvd, gd, bd, True, mom, eps)
    y_ref = act_ref(u, act)
    y_ref.backward(dy.double().reshape(-1, c))
    xg = x.cuda()
    rmg, rvg, nbt = rm.cuda(), rv.cuda(), torch.zeros(1, dtype=torch.int64).cuda()
    mean, rstd = K.bn_stats(xg, c, eps, mom, rmg, rvg, nbt)
    y = K.bn_act_fwd(xg, c, mean, rstd, gamma.cuda(), beta.cuda(), act)
    assert rel(y.reshape(-1, c), y_ref) < 2e-5
    assert rel(rmg, rmd) < 1e-5 and rel(rvg, rvd) < 1e-5 and int(nbt.item()) == 1
    dgam, dbet = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    dx = K.bn_act_bwd(xg, dy.cuda(), c, mean, rstd, gamma.cuda(), beta.cuda(), act, True, dgam, dbet, False)
    assert rel(dx.reshape(-1, c), xd.grad) < 5e-5
    assert rel(dgam, gd.grad) < 5e-5 and rel(dbet, bd.grad) < 5e-5
    # eval mode: running statistics, constants in backward
    xe = x.double().reshape(-1, c).requires_grad_(True)
    ye = act_ref(F.batch_norm(xe, rmd, rvd, gamma.double(), beta.double(), False, mom, eps), act)
    ye.backward(dy.double().reshape(-1, c))
    m2, r2 = K.bn_eval_stats(rmg, rvg, eps)
    y2 = K.bn_act_fwd(xg, c, m2, r2, gamma.cuda(), beta.cuda(), act)
    assert rel(y2.reshape(-1, c), ye) < 2e-5
    dx2 = K.bn_act_bwd(xg, dy.cuda(), c, m2, r2, gamma.cuda(), beta.cuda(), act, False)
    assert rel(dx2.reshape(-1, c), xe.grad) < 5e-5


@pytest.mark.parametrize("shape,act,masked", [((6, 7, 7, 256), "none", True), ((5, 14, 14, 24), "silu", True), ((4, 3, 3, 10), "silu", False)])
def test_bn_skip_stochastic_depth(K, shape, act, masked):
    """y = skip + rowscale[n] * act(BN_train(x)) and its backward (the residual block tail in one pass)"""
    g = torch.Generator().manual_seed(5)
    n, c = shape[0], shape[-1]
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    skip = torch.randn(shape, generator=g).double()
    gamma = (torch.rand(c, generator=g) + 0.5).double().requires_grad_(True)
    beta = torch.randn(c, generator=g).double().requires_grad_(True)
    rs = ((torch.rand(n, generator=g) > 0.3).double() / 0.7) if masked else None
    eps = 1e-3
    xn = x.permute(0, 3, 1, 2)
    u = F.batch_norm(xn, None, None, gamma, beta, True, 0.1, eps)
    u = F.silu(u) if act == "silu" else u
    if rs is not None:
        u = u * rs.view(n, 1, 1, 1)
    y_ref = skip.permute(0, 3, 1, 2) + u
    dy = torch.randn(shape, generator=g)
    y_ref.backward(dy.permute(0, 3, 1, 2).double())
    xg = x.detach().float().cuda()
    mean, rstd = K.bn_stats(xg, c, eps, 0.1)
    rsg = rs.float().cuda() if rs is not None else None
    hw = shape[1] * shape[2]
    y = K.bn_act_fwd(xg, c, mean, rstd, gamma.detach().float().cuda(), beta.detach().float().cuda(), act, skip.float().cuda(), True, rsg, hw)
    assert rel(from_nhwc(y), y_ref) < 2e-5
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    if rsg is None:
        rsg = torch.ones(n, device="cuda")
    dx = K.bn_act_bwd(xg, dy.cuda(), c, mean, rstd, gamma.detach().float().cuda(), beta.detach().float().cuda(), act, True, dg, db, False,
                      None, rsg, hw)
    assert rel(dx, x.grad) < 5e-5 and rel(dg, gamma.grad) < 2e-5 and rel(db, beta.grad) < 2e-5


def test_bn_residual_relu(K):
    g = torch.Generator().manual_seed(4)
    shape, c = (4, 6, 6, 32), 32
    x, res, dy = (torch.randn(*shape, generator=g) for _ in range(3))
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    xd = x.double().reshape(-1, c).requires_grad_(True)
    y_ref = F.relu(F.batch_norm(xd, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5) + res.double().reshape(-1, c))
    y_ref.backward(dy.double().reshape(-1, c))
    mean, rstd = K.bn_stats(x.cuda(), c, 1e-5, 0.1)
    y = K.bn_act_fwd(x.cuda(), c, mean, rstd, gamma.cuda(), beta.cuda(), "relu", res.cuda())
    assert rel(y.reshape(-1, c), y_ref) < 2e-5
    dx = K.bn_act_bwd(x.cuda(), dy.cuda(), c, mean, rstd, gamma.cuda(), beta.cuda(), "relu", True, y_out=y)
    assert rel(dx.reshape(-1, c), xd.grad) < 5e-5


@pytest.mark.parametrize("rows,c", [(98, 192), (392, 576), (10, 6), (77, 384), (5, 96), (9, 1024), (31, 260)])
def test_layernorm(K, rows, c):
    """the register-resident 16-byte kernels (c % 4 == 0, c <= 768: one, two or three vectors per lane, a partial last vector for c = 260)
    and the scalar fallback (c = 6, c = 1024); the backward with and without the fused skip-connection add"""
    g = torch.Generator().manual_seed(c)
    x, dy = torch.randn(rows, c, generator=g) * 3, torch.randn(rows, c, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y_ref = F.layer_norm(xd, (c,), gd, bd, 1e-5)
    y_ref.backward(dy.double())
    y, mean, rstd = K.layernorm_fwd(x.cuda(), c, 1e-5, gamma.cuda(), beta.cuda())
    assert rel(y, y_ref) < 2e-5
    dg, db = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    dx = K.layernorm_bwd(x.cuda(), dy.cuda(), c, mean, rstd, gamma.cuda(), dg, db)
    assert rel(dx, xd.grad) < 5e-5 and rel(dg, gd.grad) < 5e-5 and rel(db, bd.grad) < 5e-5
    skip = torch.randn(rows, c, generator=g)
    dx2 = K.layernorm_bwd(x.cuda(), dy.cuda(), c, mean, rstd, gamma.cuda(), None, None, skip=skip.cuda())
    assert torch.equal(dx2, dx + skip.cuda()) or rel(dx2, xd.grad + skip.double()) < 5e-5


@pytest.mark.parametrize("act", ["relu", "relu6", "silu", "gelu", "sigmoid"])
def test_act(K, act):
    g = torch.Generator().manual_seed(1)
    x, dy, b = torch.randn(33, 40, generator=g) * 3, torch.randn(33, 40, generator=g), torch.randn(40, generator=g)
    xd = x.double().requires_grad_(True)
    y_ref = act_ref(xd + b.double(), act)
    y_ref.backward(dy.double())
    assert rel(K.act_fwd(x.cuda(), 40, act, b.cuda()), y_ref) < 1e-5
    assert rel(K.act_bwd(x.cuda(), dy.cuda(), 40, act, b.cuda()), xd.grad) < 1e-5


def test_pool_se_residual(K):
    g = torch.Generator().manual_seed(2)
    n, h, w, c = 5, 7, 7, 1280
    x, dy = torch.randn(n, h, w, c, generator=g), torch.randn(n, h, w, c, generator=g)
    assert rel(K.avgpool_fwd(x.cuda()), x.double().mean((1, 2))) < 1e-5
    dp = torch.randn(n, c, generator=g)
    assert rel(K.avgpool_bwd(dp.cuda(), x.shape), (dp.double() / (h * w))[:, None, None, :].expand(n, h, w, c)) < 1e-5
    s = torch.rand(n, c, generator=g)
    assert rel(K.se_scale_fwd(x.cuda(), s.cuda()), x.double() * s.double()[:, None, None, :]) < 1e-6
    assert rel(K.se_scale_bwd_s(dy.cuda(), x.cuda()), (dy.double() * x.double()).sum((1, 2))) < 2e-5
    assert rel(K.se_scale_bwd_x(dy.cuda(), s.cuda(), dp.cuda()),
               dy.double() * s.double()[:, None, None, :] + dp.double()[:, None, None, :] / (h * w)) < 1e-5
    rs = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25])
    assert rel(K.add_rowscale(x.cuda(), dy.cuda(), rs.cuda()), x.double() + dy.double() * rs.double()[:, None, None, None]) < 1e-6
    assert rel(K.add_rowscale(None, dy.cuda(), rs.cuda()), dy.double() * rs.double()[:, None, None, None]) < 1e-6
    assert rel(K.add(x.cuda(), dy.cuda()), x.double() + dy.double()) < 1e-6
    assert rel(K.mul(x.cuda(), dy.cuda()), x.double() * dy.double()) < 1e-6
    ws = K.weighted_sum(x.cuda().view(-1)[:1000], dy.cuda().view(-1)[:1000], 0.5)
    assert abs(float(ws.item()) - 0.5 * float((x.view(-1)[:1000].double() * dy.view(-1)[:1000].double()).sum())) < 1e-4
    sr = K.scale_rows(dp.cuda(), s[:, 0].contiguous().cuda(), 2.0)
    assert rel(sr, dp.double() * s[:, :1].double() * 2.0) < 1e-6


def test_dropout_mask_statistics(K):
    K.rng_counter("cuda", seed=1234)
    m1 = K.dropout_mask((160, 1280), 0.8, "cuda").cpu()
    m2 = K.dropout_mask((160, 1280), 0.8, "cuda").cpu()
    vals = torch.unique(m1)
    assert len(vals) == 2 and abs(float(vals[1]) - 1.25) < 1e-6 and float(vals[0]) == 0.0
    assert abs(float((m1 > 0).float().mean()) - 0.8) < 0.01
    assert not torch.equal(m1, m2)          # the device counter advanced
    K.rng_counter("cuda", seed=1234)
    assert torch.equal(K.dropout_mask((160, 1280), 0.8, "cuda").cpu(), m1)


def test_adam_and_ema(K):
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(10007, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.clone().cuda(), torch.zeros(10007).cuda(), torch.zeros(10007).cuda()
    for step in range(1, 4):
        grad = torch.randn(10007, generator=g)
        ref.grad = grad.clone()
        opt.step()
        K.adam_step(p, grad.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step)
    assert rel(p, ref.data) < 1e-6
    e = torch.randn(10007, generator=g)
    eg = e.clone().cuda()
    K.ema_update(eg, p, 0.999)
    assert rel(eg, e.double() * 0.999 + 0.001 * p.cpu().double()) < 1e-6
    K.ema_update(eg, p, 0.0)
    assert torch.equal(eg, p)


@pytest.mark.parametrize("seed", [0, 5])
def test_strong_aug_matches_reference_fixture(K, seed):
    """CutOut+CutMix through the fused layout kernel against the reference's outputs (tests/golden/augs.npz)."""
    from oracle import pose_utils as OP
    D = np.load(os.path.join(os.path.dirname(__file__), "golden", "augs.npz"))
    shape = (6, 3, 48, 48)
    g = torch.Generator().manual_seed(100 + seed)
    img = torch.randn(*shape, generator=g)
    tag = f"seed{seed}_{shape[0]}x{shape[2]}"

    def boxes():
        x0, y0, x1, y1 = OP.draw_boxes(shape[0], shape[2], shape[3], 3, True)
        return torch.stack((x0, y0, x1, y1), -1).int().contiguous()

    torch.manual_seed(seed)
    co = boxes()
    out = K.strong_aug(img.cuda(), co.cuda(), None, None, 3, dst_nhwc=False)
    assert np.array_equal(out.cpu().numpy(), D[tag + "/cutout"])
    torch.manual_seed(seed)
    cm = boxes()
    perm = torch.randperm(shape[0]).int()
    out = K.strong_aug(img.cuda(), None, cm.cuda(), perm.cuda(), 3, dst_nhwc=False)
    assert np.array_equal(out.cpu().numpy(), D[tag + "/cutmix"])
    torch.manual_seed(seed)
    co = boxes()
    cm = boxes()
    perm = torch.randperm(shape[0]).int()
    out = K.strong_aug(img.cuda(), co.cuda(), cm.cuda(), perm.cuda(), 3, dst_nhwc=True)
    assert np.array_equal(out.cpu().permute(0, 3, 1, 2).numpy(), D[tag + "/both"])
    assert np.array_equal(K.nchw_to_nhwc(img.cuda()).cpu().numpy(), img.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("n,c,csq", [(7, 256, 16), (33, 960, 40), (5, 1536, 64)])
def test_fused_se_mlp(K, n, c, csq):
    g = torch.Generator().manual_seed(c)
    p, ds = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    w1, b1 = torch.randn(csq, c, generator=g) / c ** 0.5, torch.randn(csq, generator=g) * 0.1
    w2, b2 = torch.randn(c, csq, generator=g) / csq ** 0.5, torch.randn(c, generator=g) * 0.1
    P, W1, B1, W2, B2 = (t.double().requires_grad_(True) for t in (p, w1, b1, w2, b2))
    a1_ref = P @ W1.t() + B1
    s_ref = torch.sigmoid(F.silu(a1_ref) @ W2.t() + B2)
    s_ref.backward(ds.double())
    a1, s = K.se_mlp_fwd(p.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    assert rel(a1, a1_ref) < 1e-5 and rel(s, s_ref) < 1e-5
    grads = [torch.ones(csq, c).cuda(), torch.ones(csq).cuda(), torch.ones(c, csq).cuda(), torch.ones(c).cuda()]   # accumulate onto ones
    dp = K.se_mlp_bwd(ds.cuda(), s, a1, p.cuda(), w1.cuda(), w2.cuda(), *grads)
    assert rel(dp, P.grad) < 2e-5
    for got, ref in zip(grads, (W1.grad, B1.grad, W2.grad, B2.grad)):
        assert rel(got, ref + 1.0) < 2e-5


@pytest.mark.parametrize("seg", [None, 3])
@pytest.mark.parametrize("stride,cin,cout", [(1, 32, 32), (2, 24, 40)])
def test_mbconv_dw_bn_se_section_fused_matches_layerwise(seg, stride, cin, cout):
    """The fused depthwise-BatchNorm + SiLU + SqueezeExcitation section of an MBConv block (squeeze pass, scale pass, one
    reduction + one apply pass in the backward: suhpe_bn_act_scale_fwd / suhpe_se_bn_bwd_*) against the layer-by-layer kernels
    it replaces (bn_act_pool_fwd, se_scale_*, bn_act_bwd): outputs, input gradient, every parameter gradient and the running
    statistics, with one and with two batch segments.  The oracle comparison of the whole network is tests/test_gpu_effnet.py."""
    from semiuhpe_amd import layers as L
    from semiuhpe_amd.src import efficientnet_v2 as E
    torch.manual_seed(3)
    blk = E.MBConv(4, 3, stride, cin, cout, 0.0).cuda()
    g = torch.Generator().manual_seed(4)
    for m in blk.modules():
        if isinstance(m, L.BatchNorm2d):
            m.weight.data.copy_((torch.rand(m.num_features, generator=g) * 0.5 + 0.75).cuda())
            m.bias.data.copy_((torch.randn(m.num_features, generator=g) * 0.2).cuda())
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(8, 14, 14, cin, generator=g).cuda()
    ho = 14 // stride
    dy = torch.randn(8, ho, ho, cout, generator=g).cuda()

    def run(fused):
        E.FUSE_DW_BN_SE = fused
        blk.load_state_dict(state0)
        blk.train()
        for p in blk.parameters():
            if p.grad is not None:
                p.grad.zero_()
        tape = []
        with L.batch_segments(seg):
            y = blk.fwd(x, tape)
            dx = blk.bwd(dy, tape)
        assert not tape
        return y, dx, {n: p.grad.clone() for n, p in blk.named_parameters()}, {k: v.clone() for k, v in blk.state_dict().items() if "running" in k}

    try:
        y0, dx0, g0, r0 = run(False)
        y1, dx1, g1, r1 = run(True)
    finally:
        E.FUSE_DW_BN_SE = True
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(y1, y0) < 2e-5 and rel(dx1, dx0) < 1e-4
    for n in g0:
        assert rel(g1[n], g0[n]) < 2e-4, n
    for k in r0:
        assert rel(r1[k], r0[k]) < 1e-6, k
    # eval mode, no tape: same section with the running statistics
    blk.eval()
    E.FUSE_DW_BN_SE = False
    ye0 = blk.fwd(x, None)
    E.FUSE_DW_BN_SE = True
    ye1 = blk.fwd(x, None)
    assert rel(ye1, ye0) < 2e-5


@pytest.mark.parametrize("n,h,c,stride,seg", [(6, 14, 64, 1, None), (6, 14, 64, 1, 2), (5, 15, 48, 2, 2), (160, 14, 512, 1, 32), (40, 28, 256, 2, 8),
                                               (3, 7, 1536, 1, 1), (5, 28, 64, 1, 2), (3, 56, 96, 1, None)])
def test_dwconv_statistics_epilogue(n, h, c, stride, seg):
    """suhpe_dwconv_fwd_stats: same output bits as suhpe_dwconv_fwd, and its (S, Q, k, n) records finalise to the BatchNorm
    statistics (mean, rstd, running statistics) a separate statistics pass over the output gives -- one and two batch segments,
    the four-row stride-1 kernel (large batches) and the row kernel (small batches, stride 2)."""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(n * 100 + c)
    x = (torch.randn(n, h, h, c, generator=g) * 1.5 + 0.7).cuda()
    w = torch.randn(c, 1, 3, 3, generator=g).cuda()
    w = K.to_filter_layout(w)
    assert K.dwconv_stats_blocks(x.shape, w.shape, stride, 1) > 0
    y0 = K.conv2d_fwd(x, w, None, stride, 1, c)
    out = K.dwconv_fwd_stats(x, w, stride, 1, seg=seg)
    y = out[0]
    # same arithmetic in the same order; the two template instantiations may contract a multiply-add differently (1 ulp)
    assert float((y - y0).abs().max()) <= 2e-6 * float(y0.abs().max()), float((y - y0).abs().max())
    y0 = y
    hw = y.shape[1] * y.shape[2]
    rm0, rv0 = torch.zeros(c).cuda(), torch.ones(c).cuda()
    rm1, rv1 = rm0.clone(), rv0.clone()
    nbt0, nbt1 = torch.zeros((), dtype=torch.long).cuda(), torch.zeros((), dtype=torch.long).cuda()
    if seg is None:
        m0, r0 = K.bn_stats(y0, c, 1e-3, 0.1, rm0, rv0, nbt0)
        m1, r1 = K.bn_stats_from_partials(out[1], out[2], y.numel() // c, c, 1e-3, 0.1, rm1, rv1, nbt1)
    else:
        m0, r0 = K.bn_stats_seg(y0, seg * hw, c, 1e-3, 0.1, rm0, rv0, nbt0)
        first, boundary = out[3]
        m1, r1 = K.bn_stats_from_partials_seg(out[1], out[2], first, boundary, n * hw, seg * hw, c, 1e-3, 0.1, rm1, rv1, nbt1)
    ref = y0.double().reshape(n, hw, c)
    segs = [(0, n)] if seg is None else [(0, seg), (seg, n)]
    for i, (lo, hi) in enumerate(segs):
        mu = ref[lo:hi].reshape(-1, c).mean(0)
        var = ref[lo:hi].reshape(-1, c).var(0, unbiased=False)
        mm = m1 if seg is None else m1[i]
        rr = r1 if seg is None else r1[i]
        assert float((mm.double().cpu() - mu.cpu()).abs().max()) < 2e-6 * max(1.0, float(mu.abs().max()))
        assert float((rr.double().cpu() * (var.cpu() + 1e-3).sqrt() - 1).abs().max()) < 5e-6
    assert float((m1 - m0).abs().max()) < 2e-6 * max(1.0, float(m0.abs().max())) and float((r1 / r0 - 1).abs().max()) < 5e-6
    assert float((rm1 - rm0).abs().max()) < 1e-6 and float((rv1 / rv0 - 1).abs().max()) < 1e-5 and int(nbt1) == int(nbt0)


HALO_SHAPES = [(3, 48, 192, 1, 56, 56), (2, 24, 24, 1, 112, 112), (5, 64, 256, 1, 28, 28), (9, 128, 128, 2, 14, 14), (70, 64, 64, 1, 7, 7),
               (3, 96, 160, 1, 16, 24), (2, 192, 48, 1, 56, 56), (33, 256, 256, 2, 14, 14)]


@pytest.mark.parametrize("cfg", HALO_SHAPES, ids=str)
def test_halo_tile_kernel_3x3(K, cfg):
    """igemm_nt3x3_kernel (conv_halo.inl): every halo tile, forward and input gradient, against float64 convolutions -- 8x8 patch
    tiles (56 / 112 / 16x24 images; 24-, 32- and 48-channel chunks), run tiles with zero columns / zero rows between images
    (28, 14, 7: several images per tile, a last partial tile), grouped filters, output widths that pad the column tile; the
    BatchNorm statistics epilogue with a segment boundary and the folded eval-mode epilogue on top."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    n, cin, cout, groups, h, w = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin // groups, 3, 3, generator=g) / (3 * (cin // groups) ** 0.5)
    dy = torch.randn(n, cout, h, w, generator=g)
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, wt.double(), None, 1, 1, 1, groups)
    ref.backward(dy.double())
    xh, wh, dyh = nhwc(x), K.to_filter_layout(wt.cuda()), nhwc(dy)
    d = K.conv_desc(xh.shape, wh.shape, 1, 1, groups)
    halo_tiles = [t for t in range(lib.suhpe_conv2d_tile_count(0)) if b"nt3x3" in lib.suhpe_conv2d_tile_name(0, t)]
    assert len(halo_tiles) == 5                             # four 32-column-block tiles + the 48-column tile (conv_halo.inl)
    ran = 0
    try:
        for tile in halo_tiles:
            for pid in (0, 1):
                assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) == 0, (tile, pid)
            lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
            assert b"nt3x3" in lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 0) and b"nt3x3" in lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 1)
            y = K.conv2d_fwd(xh, wh, None, 1, 1, groups)
            assert rel(from_nhwc(y), ref.detach()) < 2e-5, ("fwd", tile)
            dx = K.conv2d_dgrad(dyh, wh, xh.shape, 1, 1, groups)
            assert rel(from_nhwc(dx), xd.grad) < 2e-5, ("dgrad", tile)
            # statistics epilogue, two segments (boundary after the first sample, or the second of >= 3)
            seg = 1 if n < 3 else 2
            out = K.conv2d_fwd_stats(xh, wh, 1, 1, groups, seg=seg)
            if b"n48" in lib.suhpe_conv2d_tile_name(0, tile):     # no statistics form: the igemm tile of the same record height runs (other k order)
                assert rel(out[0], y.cpu()) < 3e-6
            else:
                assert torch.equal(out[0], y)
            first, boundary = out[3]
            mean, rstd = K.bn_stats_from_partials_seg(out[1], out[2], first, boundary, n * h * w, seg * h * w, cout, 1e-3, 0.1)
            yd = ref.detach().permute(0, 2, 3, 1).reshape(n, h * w, cout)
            for i, (lo, hi) in enumerate(((0, seg), (seg, n))):
                mu, var = yd[lo:hi].reshape(-1, cout).mean(0), yd[lo:hi].reshape(-1, cout).var(0, unbiased=False)
                assert float((mean[i].cpu().double() - mu).abs().max()) < 1e-5 * max(1.0, float(mu.abs().max())), ("mean", tile, i)
                assert float((rstd[i].cpu().double() * (var + 1e-3).sqrt() - 1).abs().max()) < 2e-5, ("rstd", tile, i)
            # folded eval epilogue: scale / shift / SiLU / skip after the activation
            sc, sh = (torch.rand(cout, generator=g) + 0.5), torch.randn(cout, generator=g) * 0.2
            r = torch.randn(n, cout, h, w, generator=g)
            ya = K.conv2d_fwd_affine(xh, wh, sc.cuda(), sh.cuda(), "silu", nhwc(r), True, 1, 1, groups)
            ref_a = F.silu(ref.detach() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]) + r.double()
            assert rel(from_nhwc(ya), ref_a) < 2e-5, ("affine", tile)
            ran += 1
    finally:
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    assert ran == 5


@pytest.mark.parametrize("seg", [None, 3])
@pytest.mark.parametrize("kind,stride,cin,cout", [("fused", 1, 32, 32), ("fused", 2, 24, 48), ("mb", 1, 32, 32), ("mb", 2, 40, 64)])
def test_producer_batchnorm_applied_by_consumer_loader(kind, stride, cin, cout, seg):
    """suhpe_conv2d_fwd_stats_xf / suhpe_conv2d_wgrad_partial_xf: the 1x1 project conv of a FusedMBConv / MBConv block applies the
    producer's train-mode BatchNorm + SiLU (+ the SqueezeExcitation scale) while it loads, instead of reading a materialised
    activated tensor (layers.XF_FUSION): outputs, input gradient, every parameter gradient and the running statistics equal the
    unfused kernels' -- one and two batch segments (a straddling row tile included: 8 x 14 x 14 rows, boundary after 3 samples)."""
    from semiuhpe_amd import layers as L
    from semiuhpe_amd.src import efficientnet_v2 as E
    torch.manual_seed(5)
    blk = (E.FusedMBConv if kind == "fused" else E.MBConv)(4, 3, stride, cin, cout, 0.0).cuda()
    g = torch.Generator().manual_seed(6)
    for m in blk.modules():
        if isinstance(m, L.BatchNorm2d):
            m.weight.data.copy_((torch.rand(m.num_features, generator=g) * 0.5 + 0.75).cuda())
            m.bias.data.copy_((torch.randn(m.num_features, generator=g) * 0.2).cuda())
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(8, 14, 14, cin, generator=g).cuda()
    ho = 14 // stride
    dy = torch.randn(8, ho, ho, cout, generator=g).cuda()

    xf_default = L.XF_FUSION

    def run(fused):
        L.XF_FUSION = 2 if fused else 0
        blk.load_state_dict(state0)
        blk.train()
        for p in blk.parameters():
            if p.grad is not None:
                p.grad.zero_()
        tape = []
        with L.batch_segments(seg):
            y = blk.fwd(x, tape)
            seen = {}

            def walk(e):
                if torch.is_tensor(e):
                    if e.dim() == 4:
                        seen[e.data_ptr()] = e.numel()
                elif isinstance(e, (tuple, list)):
                    for q in e:
                        walk(q)
            walk(tape)
            n_saved = sum(seen.values())                    # distinct activation-sized tensors held for the backward
            dx = blk.bwd(dy, tape)
        assert not tape
        return y, dx, {n: p.grad.clone() for n, p in blk.named_parameters()}, {k: v.clone() for k, v in blk.state_dict().items() if "running" in k}, n_saved

    try:
        y0, dx0, g0, r0, saved0 = run(False)
        y1, dx1, g1, r1, saved1 = run(True)
    finally:
        L.XF_FUSION = xf_default
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(y1, y0) < 2e-5 and rel(dx1, dx0) < 1e-4
    for n in g0:
        assert rel(g1[n], g0[n]) < 2e-4, n
    for k in r0:
        assert rel(r1[k], r0[k]) < 1e-6, k
    if kind == "fused" or blk._dw_se_fused():                # (an SE block whose squeeze width is not a multiple of 4 keeps the layer-by-layer form)
        assert saved1 < saved0                               # the activated tensor is not on the tape any more


@pytest.mark.parametrize("ws,heads,B,H", [(7, 6, 6, 28), (14, 12, 5, 14), (7, 18, 9, 7)])
def test_attention_bias_gradient_fixed_order(ws, heads, B, H):
    """suhpe_window_attn_dbias: the bias-table gradient reduced without atomics -- equal to the atomic accumulation within summation
    noise, to a float64 scatter-add of dS exactly to rounding, and bit-identical from run to run"""
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src import tiny_vit as T
    torch.manual_seed(ws * heads)
    att = T.Attention(heads * 32, 32, heads, ws).cuda()
    att.attention_biases.data.normal_(0, 0.5)
    g = torch.Generator().manual_seed(B)
    qkv = torch.randn(B * H * H, heads * 96, generator=g).cuda()
    dout = torch.randn(B * H * H, heads * 32, generator=g).cuda()
    idx = att._index(qkv.device)
    noff = att.attention_biases.shape[1]
    out, probs = K.window_attn_fwd(qkv, att.attention_biases, idx, B, H, H, ws, heads, att.scale)

    def run(inverse):
        db = torch.zeros(heads, noff, device="cuda")
        dq = K.window_attn_bwd(qkv, probs, dout, idx, B, H, H, ws, heads, noff, att.scale, db, inverse=inverse)
        return dq, db

    dq_a, db_atomic = run(None)
    inv = att._inverse_index(qkv.device)
    dq1, db1 = run(inv)
    dS = K.workspace("attn", probs.numel() * 4, qkv.device)[:probs.numel() * 4].view(torch.float32).view(-1, heads, ws * ws * ws * ws).clone()
    dq2, db2 = run(inv)
    assert torch.equal(dq1, dq_a) and torch.equal(db1, db2)                 # reproducible bit for bit
    ref = torch.zeros(heads, noff, dtype=torch.float64)
    ref.index_add_(1, att.attention_bias_idxs.reshape(-1).cpu(), dS.double().sum(0).cpu())
    scale = float(ref.abs().max())
    assert float((db1.double().cpu() - ref).abs().max()) < 2e-6 * scale
    assert float((db_atomic.double().cpu() - ref).abs().max()) < 2e-5 * scale
    # accumulation semantics: a second call adds on top
    db = db1.clone()
    K.window_attn_bwd(qkv, probs, dout, idx, B, H, H, ws, heads, noff, att.scale, db, inverse=inv)
    assert float((db - 2 * db1).abs().max()) < 1e-6 * scale


@pytest.mark.parametrize("pin", [None, "igemm_nt_kernel<256,32,64,32,4>", "igemm_nt_kernel<128,128,64,64,4>", "gemm_nt_resident<64,2,8>"])
@pytest.mark.parametrize("rowscale", [False, True])
def test_input_transform_conv_on_every_reachable_tile(K, pin, rowscale):
    """a 1x1 conv with the loader's input transform (suhpe_conv2d_fwd_stats_xf) and a narrow output (cout = 32, M >= 2048)
    reaches the 256x32 tile through the heuristics; that tile once had no transform instantiation (a silent no-op).  Every tile that
    can be pinned for such a conv -- 256x32, a wide-accumulator tile, a resident tile (both without a transform form: remapped) --
    must give the result of the unfused chain, statistics records included."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    n, hw, cin, cout = 12, 14, 96, 32                       # M = 2352
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, hw, hw, cin, generator=g).cuda()
    w = K.to_filter_layout((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda())
    sc, sh = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.3).cuda()
    rs = (torch.rand(n, cin, generator=g) + 0.5).cuda() if rowscale else None
    a = torch.nn.functional.silu(x * sc + sh)
    if rowscale:
        a = a * rs[:, None, None, :]
    y_ref, part_ref, tiles_ref = K.conv2d_fwd_stats(a.contiguous(), w, 1, 0, 1)
    st_ref = K.bn_stats_from_partials(part_ref, tiles_ref, n * hw * hw, cout, 1e-3, 0.1)
    d = K.conv_desc(x.shape, w.shape, 1, 0, 1)
    names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
    try:
        if pin is not None:
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names.index(pin)) == 0
        y, part, tiles = K.conv2d_fwd_stats(x, w, 1, 0, 1, xf=(sc, sh, rs, "silu", hw * hw, 0))
        assert part.numel() == cout * tiles * 4
        st = K.bn_stats_from_partials(part, tiles, n * hw * hw, cout, 1e-3, 0.1)
    finally:
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
    assert rel(y, y_ref.cpu()) < 2e-5                        # (the loader's SiLU is v_exp / v_rcp based: ~1 ulp off torch's)
    assert rel(st[0], st_ref[0].cpu()) < 2e-5 and rel(st[1], st_ref[1].cpu()) < 2e-5


@pytest.mark.parametrize("rows,cin,cout,act", [(784, 192, 768, "gelu"), (392, 576, 2304, "gelu"), (1000, 64, 72, "silu"), (6272, 384, 1536, "gelu")])
def test_linear_activation_epilogues(K, rows, cin, cout, act):
    """suhpe_conv2d_fwd_act2 / suhpe_conv2d_dgrad_actgrad (TinyViT's fc1 -> GELU -> fc2): pre-activation, activation and the input
    gradient multiplied by act'(pre) from the GEMM epilogues, bit-equal to the unfused chain (Linear, act_fwd, Linear backward, act_bwd),
    on the heuristic tile and with persistent / resident / wide tiles pinned (all remapped to the igemm tile that carries the epilogue)"""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, 1, 1, cin, generator=g).cuda()
    w1 = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
    b1 = (torch.randn(cout, generator=g) * 0.1).cuda()
    w2 = (torch.randn(cin, cout, 1, 1, generator=g) / cout ** 0.5).cuda()
    dy = torch.randn(rows, 1, 1, cin, generator=g).cuda()                            # gradient of fc2's output
    h_ref = K.conv2d_fwd(x, w1, b1, 1, 0, 1)
    a_ref = K.act_fwd(h_ref, cout, act)
    da_ref = K.conv2d_dgrad(dy, w2, a_ref.shape, 1, 0, 1)
    dh_ref = K.act_bwd(h_ref, da_ref, cout, act)
    d1, d2 = K.conv_desc(x.shape, w1.shape, 1, 0, 1), K.conv_desc(a_ref.shape, w2.shape, 1, 0, 1)
    names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
    pins = [None, "gemm_nt_persistent<64,64,32,32>", "gemm_nt_resident<64,2,8>", "igemm_nt_kernel<128,128,64,64,4>", "igemm_nt_kernel<128,32,32,32,4>"]
    try:
        for pin in pins:
            for d, pid in ((d1, 0), (d2, 1)):
                assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1 if pin is None else names.index(pin)) == 0 or "resident" in pin
            h, a = K.conv2d_fwd_act2(x, w1, b1, act)
            dh = K.conv2d_dgrad_actgrad(dy, w2, a_ref.shape, h_ref, act)
            assert torch.equal(h, h_ref) and torch.equal(a, a_ref), pin
            assert torch.equal(dh, dh_ref), pin
    finally:
        lib.suhpe_conv2d_set_tile(ctypes.byref(d1), 0, -1)
        lib.suhpe_conv2d_set_tile(ctypes.byref(d2), 1, -1)
    ad = torch.nn.functional.gelu(h_ref.double().cpu()) if act == "gelu" else torch.nn.functional.silu(h_ref.double().cpu())
    assert rel(a, ad) < 1e-6


RESIDENT_SHAPES = [
    # (n, cin, cout, groups, h, w, seg): 1x1 / stride 1 convs as the resident kernel sees them
    (40, 160, 960, 1, 14, 14, 8),      # the C2 expand shape (M = 7840: a last partial row tile at both tile heights)
    (8, 40, 72, 1, 14, 14, 3),         # K = 40 (one padded slab), N = 72 (a ragged stripe), boundary inside a 64-row unit (3 * 196 = 588)
    (5, 256, 96, 2, 7, 7, None),       # grouped, M = 245 < one 128-row tile
    (160, 64, 256, 1, 7, 7, 32),       # boundary at a unit edge (32 * 49 = 1568 = 24.5 units: straddles)
    (3, 608, 64, 1, 9, 9, None),       # the largest K of the 64-row tile (the 128-row tile is refused)
]


@pytest.mark.parametrize("cfg", RESIDENT_SHAPES, ids=str)
def test_resident_pointwise_kernel(K, cfg):
    """gemm_nt_resident (conv_resident.inl): both resident tiles, forward and input gradient, are bit-equal to the igemm tile (same k
    order per output element) -- plain, with bias, with the skip add, with the folded eval BatchNorm epilogue -- and their per-unit
    BatchNorm statistics records (one or two batch segments, a unit that straddles the boundary) finish to the statistics of the
    stand-alone pass; refused where the BM x K block does not fit the LDS."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    n, cin, cout, groups, h, w, seg = cfg
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(n, h, w, cin, generator=g) + 0.2).cuda()
    wg = K.to_filter_layout((torch.randn(cout, cin // groups, 1, 1, generator=g) / (cin // groups) ** 0.5).cuda())
    bias = torch.randn(cout, generator=g).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    skip = torch.randn(n, h, w, cin, generator=g).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    res = torch.randn(n, h, w, cout, generator=g).cuda()
    d = K.conv_desc(x.shape, wg.shape, 1, 0, groups)
    tiles = [t for t in range(lib.suhpe_conv2d_tile_count(0)) if b"resident" in lib.suhpe_conv2d_tile_name(0, t)]
    assert len(tiles) == 3
    igemm = [t for t in range(lib.suhpe_conv2d_tile_count(0)) if lib.suhpe_conv2d_tile_name(0, t) == b"igemm_nt_kernel<64,64,32,32,4>"][0]
    rows = n * h * w

    def run_all():
        out = [K.conv2d_fwd(x, wg, None, 1, 0, groups), K.conv2d_fwd(x, wg, bias, 1, 0, groups),
               K.conv2d_dgrad(dy, wg, x.shape, 1, 0, groups), K.conv2d_dgrad(dy, wg, x.shape, 1, 0, groups, skip=skip),
               K.conv2d_fwd_affine(x, wg, scale, shift, "silu", res, True, 1, 0, groups),
               K.conv2d_fwd_affine(x, wg, scale, shift, "relu", res, False, 1, 0, groups)]
        if seg is None:
            y, part, nt = K.conv2d_fwd_stats(x, wg, 1, 0, groups)
            st = K.bn_stats_from_partials(part, nt, rows, cout, 1e-3, 0.1)
        else:
            y, part, nt, (first, extra) = K.conv2d_fwd_stats(x, wg, 1, 0, groups, seg=seg)
            st = K.bn_stats_from_partials_seg(part, nt, first, extra, rows, seg * h * w, cout, 1e-3, 0.1)
        return out + [y], st

    try:
        for pid in (0, 1):
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, igemm) == 0
        ref, st_ref = run_all()
        ran = 0
        for tile in tiles:
            ok = [lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) == 0 for pid in (0, 1)]
            bm = 128 if b"<128" in lib.suhpe_conv2d_tile_name(0, tile) else 64
            fits = lambda kk: bm * ((kk + 31) // 32 * 32 + 4) * 4 <= 160 * 1024      # the BM x K block (rows padded by 4 floats) in 160 KB of LDS
            assert ok[0] == fits(cin // groups) and ok[1] == fits(cout // groups), (tile, ok)
            for pid in (0, 1):
                if not ok[pid]:
                    lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, igemm)
            lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
            assert (b"resident" in lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 0)) == ok[0]
            out, st = run_all()
            for i, (a, b) in enumerate(zip(out, ref)):
                assert torch.equal(a, b), (tile, i)
            for a, b in zip(st, st_ref):
                assert rel(a, b.cpu()) < 2e-6, tile
            ran += ok[0]
        assert ran >= 1
        yd = ref[0].double().cpu().reshape(rows, cout)
        if seg is None:
            assert rel(st_ref[0], yd.mean(0)) < 2e-5 and rel(st_ref[1], (yd.var(0, unbiased=False) + 1e-3).rsqrt()) < 2e-5
        else:
            ra = seg * h * w
            for i, part_rows in enumerate((yd[:ra], yd[ra:])):
                assert rel(st_ref[0][i], part_rows.mean(0)) < 2e-5 and rel(st_ref[1][i], (part_rows.var(0, unbiased=False) + 1e-3).rsqrt()) < 2e-5
    finally:
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)


DMA_SHAPES = [
    # (n, cin, cout, h, w, seg): plain 1x1 / stride 1 convs with K % 32 == 0 as gemm_nt_dma sees them
    (40, 160, 960, 14, 14, 8),         # the C2 expand shape at a quarter of the batch: N = 960 = 7.5 column tiles, M = 7840 = 61.25 row tiles (XCD padding tiles skipped)
    (160, 256, 1536, 7, 7, 32),        # the 7 x 7 expand shape, boundary inside a 64-row unit (32 * 49 = 1568 = 24.5 units)
    (7, 32, 36, 5, 5, 2),              # ONE k-slab per tile (a tile ends at every slab: the store-count side of the waits), N < one stripe, M = 175
    (24, 64, 200, 14, 14, None),       # two slabs per tile, ragged columns (200 = 1.56 column tiles)
    (2, 96, 640, 3, 3, 1),             # fewer tiles than workgroups (M = 18)
    (300, 128, 512, 14, 14, 60),       # long streams: 9 tiles per workgroup
]


@pytest.mark.parametrize("cfg", DMA_SHAPES, ids=str)
def test_lds_dma_pointwise_kernel(K, cfg):
    """gemm_nt_dma (conv_dma.inl, round 5): forward and input gradient are bit-equal to the igemm tile (same k order per output element), its
    per-unit BatchNorm statistics records (one or two batch segments, a unit that straddles the boundary) finish to the statistics of the igemm
    tile's records and of a float64 reference; forms it lacks (bias, skip add, folded eval epilogue) fall back to a direct tile and stay bit-equal;
    repeated launches give the same bits (a race in the hand-counted waits would show here)."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
    n, cin, cout, h, w, seg = cfg
    g = torch.Generator().manual_seed(47)
    x = (torch.randn(n, h, w, cin, generator=g) + 0.2).cuda()
    wg = K.to_filter_layout((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda())
    bias = torch.randn(cout, generator=g).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    skip = torch.randn(n, h, w, cin, generator=g).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    res = torch.randn(n, h, w, cout, generator=g).cuda()
    d = K.conv_desc(x.shape, wg.shape, 1, 0, 1)
    names = [lib.suhpe_conv2d_tile_name(0, t) for t in range(lib.suhpe_conv2d_tile_count(0))]
    dma = names.index(b"gemm_nt_dma<128,128>")
    igemm = names.index(b"igemm_nt_kernel<64,64,32,32,4>")
    rows = n * h * w

    def run_all():
        out = [K.conv2d_fwd(x, wg, None, 1, 0, 1), K.conv2d_dgrad(dy, wg, x.shape, 1, 0, 1),
               K.conv2d_fwd(x, wg, bias, 1, 0, 1), K.conv2d_dgrad(dy, wg, x.shape, 1, 0, 1, skip=skip),
               K.conv2d_fwd_affine(x, wg, scale, shift, "silu", res, True, 1, 0, 1)]
        if seg is None:
            y, part, nt = K.conv2d_fwd_stats(x, wg, 1, 0, 1)
            st = K.bn_stats_from_partials(part, nt, rows, cout, 1e-3, 0.1, consume=False)
        else:
            y, part, nt, (first, extra) = K.conv2d_fwd_stats(x, wg, 1, 0, 1, seg=seg)
            st = K.bn_stats_from_partials_seg(part, nt, first, extra, rows, seg * h * w, cout, 1e-3, 0.1, consume=False)
        return out + [y], st

    try:
        for pid in (0, 1):
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, igemm) == 0
        ref, st_ref = run_all()
        ok = [lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, dma) == 0 for pid in (0, 1)]
        assert ok[0] == (cin % 32 == 0) and ok[1] == (cout % 32 == 0), ok
        for pid in (0, 1):
            if not ok[pid]:
                lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, igemm)
        assert (b"gemm_nt_dma" in lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 0)) == ok[0]
        for rep in range(3):
            out, st = run_all()
            for i, (a, b) in enumerate(zip(out, ref)):
                assert torch.equal(a, b), (rep, i)
            for a, b in zip(st, st_ref):
                assert rel(a, b.cpu()) < 2e-6
        yd = ref[0].double().cpu().reshape(rows, cout)
        if seg is None:
            assert rel(st[0], yd.mean(0)) < 2e-5 and rel(st[1], (yd.var(0, unbiased=False) + 1e-3).rsqrt()) < 2e-5
        else:
            ra = seg * h * w
            for i, part_rows in enumerate((yd[:ra], yd[ra:])):
                assert rel(st[0][i], part_rows.mean(0)) < 2e-5 and rel(st[1][i], (part_rows.var(0, unbiased=False) + 1e-3).rsqrt()) < 2e-5
    finally:
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)


NARROW_SHAPES = [
    # (n, cin, cout, h, w, seg, act): FusedMBConv project convs as the narrow LDS-DMA tile sees them (outputs of at most 64 channels)
    (20, 192, 48, 28, 28, 4, "silu"),      # the C2 stage-2 shape at an eighth of the rows: N = 48 (16 dropped columns), boundary at a 32-row unit edge
    (9, 256, 64, 14, 14, 2, "silu"),       # N = 64 exactly, boundary inside a unit (2 * 196 = 392 = 12.25 units) and inside a 256-row tile
    (5, 96, 48, 10, 10, None, "relu"),     # three slabs, ReLU, one segment, M = 500 (a last partial tile)
    (3, 32, 12, 6, 6, 1, "none"),          # ONE slab per tile, identity activation, 12 outputs
    (130, 64, 24, 14, 14, 26, "silu"),     # long streams (100 row tiles), two slabs per tile
]


@pytest.mark.parametrize("cfg", NARROW_SHAPES, ids=str)
def test_lds_dma_narrow_kernel_with_loader_transform(K, cfg):
    """gemm_nt_dma<256,64> (conv_dma.inl): forward with the producer's BatchNorm + activation applied to the MFMA fragments (XF = 1) equals the igemm
    tile's XF form (the same fmaf + activation per element; the k order is the same: bit-equal or within fp32 rounding of the transcendental), its
    32-row statistics records finish to the igemm tile's statistics, the plain form and the input gradient (N = cin <= 64) are bit-equal, repeated
    launches give the same bits, and against float64 the result is as close as the igemm tile's."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    n, cin, cout, h, w, seg, act = cfg
    g = torch.Generator().manual_seed(53)
    x = (torch.randn(n, h, w, cin, generator=g) * 1.3 + 0.2).cuda()
    wg = K.to_filter_layout((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda())
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    nseg = 2 if seg else 1
    scale, shift = (torch.rand(nseg, cin, generator=g) + 0.5).cuda(), (torch.randn(nseg, cin, generator=g) * 0.3).cuda()
    d = K.conv_desc(x.shape, wg.shape, 1, 0, 1)
    names = [lib.suhpe_conv2d_tile_name(0, t) for t in range(lib.suhpe_conv2d_tile_count(0))]
    narrow, igemm = names.index(b"gemm_nt_dma<256,64>"), names.index(b"igemm_nt_kernel<64,64,32,32,4>")
    rows, hw = n * h * w, h * w
    rows_a = seg * hw if seg else 0
    xf = (scale, shift, None, act, hw, rows_a)

    def run_all():
        if seg is None:
            y, part, nt = K.conv2d_fwd_stats(x, wg, 1, 0, 1, xf=xf)
            st = K.bn_stats_from_partials(part, nt, rows, cout, 1e-3, 0.1, consume=False)
            y0, part0, nt0 = K.conv2d_fwd_stats(x, wg, 1, 0, 1)
            st0 = K.bn_stats_from_partials(part0, nt0, rows, cout, 1e-3, 0.1, consume=False)
        else:
            y, part, nt, (first, extra) = K.conv2d_fwd_stats(x, wg, 1, 0, 1, seg=seg, xf=xf)
            st = K.bn_stats_from_partials_seg(part, nt, first, extra, rows, rows_a, cout, 1e-3, 0.1, consume=False)
            y0, part0, nt0, (first0, extra0) = K.conv2d_fwd_stats(x, wg, 1, 0, 1, seg=seg)
            st0 = K.bn_stats_from_partials_seg(part0, nt0, first0, extra0, rows, rows_a, cout, 1e-3, 0.1, consume=False)
        return [y, y0, K.conv2d_fwd(x, wg, None, 1, 0, 1)], [st, st0]

    try:
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, igemm) == 0
        ref, st_ref = run_all()
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, narrow) == 0
        assert lib.suhpe_conv2d_fwd_stats_tile_rows(ctypes.byref(d)) == 32
        for rep in range(3):
            out, st = run_all()
            assert rel(out[0], ref[0].cpu()) < 2e-6, rep                     # XF form (hardware exp / rcp: allow the last bit of the activation)
            assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2]), rep
            for a, b in zip(st, st_ref):
                for u, v in zip(a, b):
                    assert rel(u, v.cpu()) < 5e-6
            if rep:
                assert torch.equal(out[0], first_out)                        # run-to-run bit-identical
            first_out = out[0]
        # float64 reference of the transformed GEMM
        xs = x.double().cpu().reshape(rows, cin)
        seg_of = (torch.arange(rows) >= rows_a).long() if seg else torch.zeros(rows, dtype=torch.long)
        u = xs * scale.double().cpu()[seg_of] + shift.double().cpu()[seg_of]
        a64 = u * torch.sigmoid(u) if act == "silu" else (u.clamp(min=0) if act == "relu" else u)
        y64 = a64 @ wg.double().cpu().reshape(cout, cin).T
        e_new, e_old = rel(out[0].reshape(rows, cout), y64), rel(ref[0].reshape(rows, cout), y64)
        assert e_new < 2e-5 and e_new < 2 * e_old + 1e-6, (e_new, e_old)
        # input gradient through the narrow tile where N = cin <= 64
        if cin <= 64 and cout % 32 == 0:
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, igemm)
            dref = K.conv2d_dgrad(dy, wg, x.shape, 1, 0, 1)
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, narrow) == 0
            assert torch.equal(K.conv2d_dgrad(dy, wg, x.shape, 1, 0, 1), dref)
    finally:
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)


def test_lds_dma_tile_is_refused_where_it_cannot_run(K):
    """the LDS-DMA tile needs a plain 1x1 / stride 1 conv, one group, reduction a multiple of 32: set_tile refuses everything else"""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    dma = [lib.suhpe_conv2d_tile_name(0, t) for t in range(lib.suhpe_conv2d_tile_count(0))].index(b"gemm_nt_dma<128,128>")
    for shape, wshape, stride, pad, groups, want in (((4, 8, 8, 64), (96, 64, 1, 1), 1, 0, 1, (0, 0)), ((4, 8, 8, 40), (96, 40, 1, 1), 1, 0, 1, (-1, 0)),
                                                      ((4, 8, 8, 64), (96, 32, 1, 1), 1, 0, 2, (-1, -1)), ((4, 8, 8, 64), (96, 64, 3, 3), 1, 1, 1, (-1, -1)),
                                                      ((4, 8, 8, 64), (100, 64, 1, 1), 1, 0, 1, (0, -1))):
        d = K.conv_desc(shape, wshape, stride, pad, groups)
        got = tuple(lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, dma) for pid in (0, 1))
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
        assert got == want, (shape, wshape, groups, got)


@pytest.mark.parametrize("n,h,c,seg,act", [(160, 14, 512, 32, "silu"), (160, 7, 1536, None, "silu"), (160, 28, 256, 8, "relu"), (80, 14, 960, 5, "silu"),
                                           (160, 56, 128, 32, "gelu"), (9, 14, 192, None, "gelu")])
def test_depthwise_dgrad_reduces_producer_batchnorm_sums(K, n, h, c, seg, act):
    """suhpe_dwconv_dgrad_bnsums: the depthwise input gradient is bit-equal to suhpe_dwconv_dgrad's, and the BatchNorm backward built from
    the sums it reduces (one or two batch segments) + the apply pass equals the stand-alone three-pass BatchNorm backward (same dx up to the
    summation order of the per-channel sums; dgamma / dbeta accumulated on top of what the buffers held)"""
    g = torch.Generator().manual_seed(17)
    x1 = (torch.randn(n, h, h, c, generator=g) * 1.5 + 0.3).cuda()                  # the BatchNorm's input (expand conv output)
    dy = torch.randn(n, h, h, c, generator=g).cuda()                                 # gradient of the depthwise conv's output
    w = K.to_filter_layout((torch.randn(c, 1, 3, 3, generator=g) / 3).cuda())
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.2).cuda()
    assert K.dwconv_dgrad_bnsums_ok(x1.shape, w.shape, 1, 1)
    rows, hw = n * h * h, h * h
    if seg is None:
        mean, rstd = K.bn_stats(x1, c, 1e-3, 0.1)
    else:
        mean, rstd = K.bn_stats_seg(x1, seg * hw, c, 1e-3, 0.1)
    d_ref = K.conv2d_dgrad(dy, w, x1.shape, 1, 1, c)
    dg_ref, db_ref = torch.full((c,), 0.5, device="cuda"), torch.full((c,), -0.25, device="cuda")
    dg, db = dg_ref.clone(), db_ref.clone()
    if seg is None:
        dz_ref = K.bn_act_bwd(x1, d_ref, c, mean, rstd, gamma, beta, act, True, dg_ref, db_ref, True)
    else:
        dz_ref = K.bn_act_bwd_seg(x1, d_ref, seg * hw, c, mean, rstd, gamma, beta, act, dg_ref, db_ref, True)
    d, sums = K.dwconv_dgrad_bnsums(dy, w, x1.shape, 1, 1, x1, mean, rstd, gamma, beta, act, seg or 0, dg, db)
    assert torch.equal(d, d_ref)
    dz = (K.bn_act_bwd_apply(x1, d, c, mean, rstd, gamma, beta, act, sums, rows) if seg is None
          else K.bn_act_bwd_apply_seg(x1, d, seg * hw, c, mean, rstd, gamma, beta, act, sums))
    assert rel(dz, dz_ref.cpu()) < 2e-6
    assert rel(dg, dg_ref.cpu()) < 2e-5 and rel(db, db_ref.cpu()) < 2e-5
    # float64 reference of the sums themselves
    xs = x1.double().cpu().reshape(rows, c)
    parts = [(0, rows)] if seg is None else [(0, seg * hw), (seg * hw, rows)]
    m64, r64 = mean.double().cpu().reshape(-1, c), rstd.double().cpu().reshape(-1, c)
    for i, (lo, hi) in enumerate(parts):
        xh = (xs[lo:hi] - m64[i]) * r64[i]
        u = xh * gamma.double().cpu() + beta.double().cpu()
        if act == "silu":
            ga = torch.sigmoid(u) * (1 + u * (1 - torch.sigmoid(u)))
        elif act == "gelu":
            ga = 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5
        else:
            ga = (u > 0).double()
        gg = d_ref.double().cpu().reshape(rows, c)[lo:hi] * ga
        s = sums.reshape(-1, c, 2)[i].double().cpu()
        assert rel(s[:, 0], gg.sum(0)) < 2e-5 and rel(s[:, 1], (gg * xh).sum(0)) < 2e-5


@pytest.mark.parametrize("n,h,c,seg,act,section", [(160, 14, 960, 32, "silu", True), (160, 7, 1536, 32, "silu", True), (9, 14, 192, None, "silu", True),
                                                   (7, 7, 40, 3, "silu", True), (5, 7, 96, None, "relu", True), (12, 14, 512, 4, "silu", False),
                                                   (11, 7, 264, 5, "gelu", False)])
def test_depthwise_whole_backward_in_one_launch(K, n, h, c, seg, act, section):
    """suhpe_dwconv_bwd_fused (round 6) against the three kernels it replaces: the depthwise BatchNorm + SqueezeExcitation section's apply pass
    (se_bn_bwd_apply -> dy), the filter gradient of the conv whose input is act(BN_e(e0)) (dwconv_bnact_wgrad) and the input gradient with the
    expand BatchNorm's backward sums (dwconv_dgrad_bnsums) -- one and two batch segments (the boundary inside a two-image group of 7 x 7 maps),
    ragged batches, channel counts off the 64-channel chunks; section=False: dy given (the fused kernel's plain form).  Tolerances: the filter
    gradient is summed in another order (workgroup records of other image groups), the conv's input z = act(BN_e(e0)) shares its sigmoid with act'."""
    g = torch.Generator().manual_seed(23 + n + c)
    hw, rows = h * h, n * h * h
    e0 = (torch.randn(n, h, h, c, generator=g) * 1.5 + 0.3).cuda()                  # expand conv output = input of BatchNorm_e
    y = (torch.randn(n, h, h, c, generator=g) * 0.8 - 0.1).cuda()                    # depthwise conv output = input of BatchNorm_d
    dout = torch.randn(n, h, h, c, generator=g).cuda()
    w = K.to_filter_layout((torch.randn(c, 1, 3, 3, generator=g) / 3).cuda())
    ga_e, be_e = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.2).cuda()
    ga_d, be_d = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.2).cuda()
    s_ = torch.rand(n, c, generator=g).cuda()
    dp = (torch.randn(n, c, generator=g) * 0.3).cuda()
    assert K.dwconv_bwd_fused_ok(e0.shape, w.shape, 1, 1)
    rows_a = (seg or 0) * hw
    st = (lambda t: K.bn_stats_seg(t, rows_a, c, 1e-3, 0.1)) if seg else (lambda t: K.bn_stats(t, c, 1e-3, 0.1))
    (mean_e, rstd_e), (mean_d, rstd_d) = st(e0), st(y)
    if section:
        sums5 = K.se_bn_bwd_reduce(y, dout, rows_a, c, mean_d, rstd_d, ga_d, be_d, act, hw)
        chan = K.se_bn_bwd_combine(sums5, s_, dp, rows_a // hw, hw, torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), True)
        dy = K.se_bn_bwd_apply(y, dout, rows_a, c, mean_d, rstd_d, ga_d, be_d, act, s_, dp, chan, True, hw)
        mid = (y, s_, dp, chan, mean_d, rstd_d, ga_d, be_d, act, True)
    else:
        dy, mid = dout, None
    # the three-kernel path
    dw_ref = torch.full(tuple(w.shape), 0.125, device="cuda")
    dw_ref = K.dwconv_bnact_wgrad(e0, mean_e, rstd_e, ga_e, be_e, act, dy, tuple(w.shape), 1, 1, out=dw_ref, accumulate=True, seg=seg)
    dg_ref, db_ref = torch.full((c,), 0.5, device="cuda"), torch.full((c,), -0.25, device="cuda")
    d_ref, sums_ref = K.dwconv_dgrad_bnsums(dy, w, e0.shape, 1, 1, e0, mean_e, rstd_e, ga_e, be_e, act, seg or 0, dg_ref, db_ref)
    # one launch
    dw_got = torch.full(tuple(w.shape), 0.125, device="cuda")
    dg, db = torch.full((c,), 0.5, device="cuda"), torch.full((c,), -0.25, device="cuda")
    d, sums = K.dwconv_bwd_fused(dout if section else dy, w, e0.shape, 1, 1, e0, mean_e, rstd_e, ga_e, be_e, act, seg or 0, dg, db, dw_got, mid=mid)
    assert rel(d, d_ref.cpu()) < (2e-6 if section else 1e-7)         # (plain form: the same window sums on the same dy)
    assert rel(sums, sums_ref.cpu()) < 2e-5
    assert rel(dg, dg_ref.cpu()) < 2e-5 and rel(db, db_ref.cpu()) < 2e-5
    assert rel(dw_got, dw_ref.cpu()) < 2e-5
    # ... and the filter gradient against float64 (torch): dW[c][r][s] = sum dy(p) z(p + (r - 1, s - 1)), z = act(BN_e(e0))
    m64 = mean_e.double().cpu().reshape(-1, c); r64 = rstd_e.double().cpu().reshape(-1, c)
    xs = e0.double().cpu().reshape(rows, c)
    parts = [(0, rows)] if not seg else [(0, rows_a), (rows_a, rows)]
    u = torch.cat([((xs[lo:hi] - m64[i]) * r64[i]) * ga_e.double().cpu() + be_e.double().cpu() for i, (lo, hi) in enumerate(parts)])
    z = {"silu": torch.nn.functional.silu, "relu": torch.relu, "gelu": lambda t: torch.nn.functional.gelu(t)}[act](u).reshape(n, h, h, c)
    zp = torch.nn.functional.pad(z.permute(0, 3, 1, 2), (1, 1, 1, 1))
    dy64 = dy.double().cpu().permute(0, 3, 1, 2)
    want = torch.stack([torch.stack([(dy64 * zp[:, :, r:r + h, q:q + h]).sum((0, 2, 3)) for q in range(3)], -1) for r in range(3)], -2)    # [c][3][3]
    assert rel(dw_got.reshape(c, 3, 3) - 0.125, want) < 2e-5


def _db_entries(max_elems=160 * 112 * 112 * 96):
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semiuhpe_amd", "tiles_mi355x.json")
    tiles = json.load(open(path))["tiles"]
    out = []
    for key, name in sorted(tiles.items()):
        pid, n, h, w, cin, cout, r, s_, stride, pad, groups = (int(v) for v in key.split(","))
        ho = (h + 2 * pad - r) // stride + 1
        if n not in (128, 160) or max(n * h * w * cin, n * ho * ho * cout) > max_elems:
            continue
        out.append((key, name))
    return out


def test_tile_database_is_the_tested_tile_set(K):
    """The benchmark runs with the winners of semiuhpe_amd/tiles_mi355x.json pinned; the parity tests run on the built-in
    heuristics (conftest: SUHPE_AUTOTUNE=0).  This closes the gap: every database entry of the C2 batch sizes (128 teacher /
    160 student images) is pinned by name and its forward / input gradient / filter gradient compared with the heuristic tile's
    on the same operands -- bit-equal for the igemm / persistent tiles (the k order of an output element does not depend on the
    tile), within summation-order noise for the halo tiles (chunk-major reduction) and the split-K filter gradients."""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    names = {pid: [lib.suhpe_conv2d_tile_name(pid, i).decode() for i in range(lib.suhpe_conv2d_tile_count(pid))] for pid in (0, 1, 2)}
    entries = _db_entries()
    assert len(entries) > 150
    g = torch.Generator().manual_seed(3)
    cache = {}
    checked = {0: 0, 1: 0, 2: 0}
    for key, name in entries:
        pid, n, h, w, cin, cout, r, s_, stride, pad, groups = (int(v) for v in key.split(","))
        shape_key = (n, h, w, cin, cout, r, stride, pad, groups)
        if shape_key not in cache:
            cache.clear()                                   # one shape's operands at a time
            x = torch.randn(n, h, w, cin, generator=g).cuda()
            wt = K.to_filter_layout((torch.randn(cout, cin // groups, r, r, generator=g) / (r * (cin // groups) ** 0.5)).cuda())
            ho = (h + 2 * pad - r) // stride + 1
            dy = torch.randn(n, ho, ho, cout, generator=g).cuda()
            cache[shape_key] = (x, wt, dy)
        x, wt, dy = cache[shape_key]
        d = K.conv_desc(x.shape, wt.shape, stride, pad, groups)
        run = {0: lambda: K.conv2d_fwd(x, wt, None, stride, pad, groups), 1: lambda: K.conv2d_dgrad(dy, wt, x.shape, stride, pad, groups),
               2: lambda: K.conv2d_wgrad(x, dy, wt.shape, stride, pad, groups)}[pid]
        assert name in names[pid], (key, name)
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
        lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
        default_name = lib.suhpe_conv2d_kernel_name(ctypes.byref(d), pid).decode()
        ref = run()
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, names[pid].index(name)) == 0, (key, name)
        try:
            out = run()
        finally:
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
        if pid < 2 and "nt3x3" not in name and "nt3x3" not in default_name and "wino" not in name:
            assert torch.equal(out, ref), (key, name)
        else:
            assert rel(out, ref.cpu()) < 3e-5, (key, name)
        checked[pid] += 1
    assert all(v > 30 for v in checked.values()), checked


@pytest.mark.parametrize("cfg", [(6, 48, 192, 3, 1, 1, 1, 28, 28), (40, 160, 960, 1, 1, 0, 1, 14, 14), (4, 64, 128, 3, 2, 1, 1, 30, 30), (2, 128, 128, 3, 1, 1, 2, 14, 14),
                                 (160, 960, 160, 1, 1, 0, 1, 14, 14)])
def test_dgrad_with_skip_add(K, cfg):
    """suhpe_conv2d_dgrad_add: dx = dgrad(dy) + skip from the GEMM epilogue == the input gradient followed by a separate add, bit
    for bit, on every tile of the menu (igemm, persistent, halo; stride 2 = the parity-class launches; groups)"""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    n, cin, cout, k, stride, pad, groups, h, w = cfg
    g = torch.Generator().manual_seed(21)
    wg = K.to_filter_layout((torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)).cuda())
    ho = (h + 2 * pad - k) // stride + 1
    dy = torch.randn(n, ho, ho, cout, generator=g).cuda()
    skip = torch.randn(n, h, w, cin, generator=g).cuda()
    d = K.conv_desc((n, h, w, cin), wg.shape, stride, pad, groups)
    try:
        for tile in range(-1, lib.suhpe_conv2d_tile_count(1)):
            if lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, tile) != 0:
                continue
            plain = K.conv2d_dgrad(dy, wg, (n, h, w, cin), stride, pad, groups)
            fused = K.conv2d_dgrad(dy, wg, (n, h, w, cin), stride, pad, groups, skip=skip)
            assert torch.equal(fused, plain + skip), tile
    finally:
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, -1)


@pytest.mark.parametrize("n,cout,hw,stride,seg", [(5, 24, 40, 2, None), (6, 24, 31, 2, 2), (3, 48, 28, 2, 1), (2, 48, 33, 2, None), (4, 24, 16, 1, 3), (40, 24, 30, 2, 13)])
def test_rgb_stem_direct_conv(K, n, cout, hw, stride, seg):
    """rgb_conv3x3_kernel (conv_stem.inl): the backbones' 3-channel stem conv as a direct kernel == a float64 reference to fp32 noise, with bias, and with the BatchNorm statistics records / batch segments of the GEMM epilogue's format"""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
    g = torch.Generator().manual_seed(n + cout)
    x = torch.randn(n, hw, hw, 3, generator=g).cuda()
    wt = (torch.randn(cout, 3, 3, 3, generator=g) / 5).cuda()
    wg = K.to_filter_layout(wt)
    bias = torch.randn(cout, generator=g).cuda()
    d = K.conv_desc(x.shape, wg.shape, stride, 1, 1)
    assert b"rgb_conv3x3" in lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 0)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), stride, 1).permute(0, 2, 3, 1)
    y = K.conv2d_fwd(x, wg, bias, stride, 1, 1)
    assert rel(y, ref.cpu()) < 2e-6
    res = K.conv2d_fwd_stats(x, wg, stride, 1, 1, seg=seg)
    y0 = res[0]
    rows = y0.numel() // cout
    assert rel(y0, (ref - bias.double()).cpu()) < 2e-6
    if seg:
        rows_a = seg * y0.shape[1] * y0.shape[2]
        m2, r2 = K.bn_stats_from_partials_seg(res[1], res[2], res[3][0], res[3][1], rows, rows_a, cout, 1e-3, 0.1)
        m0, r0 = K.bn_stats_seg(y0, rows_a, cout, 1e-3, 0.1)
    else:
        m2, r2 = K.bn_stats_from_partials(res[1], res[2], rows, cout, 1e-3, 0.1)
        m0, r0 = K.bn_stats(y0, cout, 1e-3, 0.1)
    assert (m2 - m0).abs().max().item() < 2e-6 and rel(r2, r0.cpu()) < 2e-6


@pytest.mark.parametrize("n,h,c", [(3, 7, 20), (5, 14, 96), (33, 7, 1536), (9, 14, 960), (2, 14, 64), (1, 7, 4)])
def test_depthwise_image_tile_kernel_against_torch(K, n, h, c):
    """dw3x3_tile_kernel (3x3 / stride 1 / pad 1 on 14 x 14 and 7 x 7 maps: whole images staged in LDS, persistent workgroups over image
    groups): forward, input gradient and filter gradient (dw3x3_wgrad_tile_kernel) against torch's fp64 depthwise convolution -- partial last groups (odd n on 7 x 7 maps), channel
    counts that are not multiples of the 64-channel chunk, more groups than resident workgroups"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, c, h, h, generator=g, dtype=torch.float64)
    w = torch.randn(c, 1, 3, 3, generator=g, dtype=torch.float64) / 3
    dy = torch.randn(n, c, h, h, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, w, None, 1, 1, 1, c).permute(0, 2, 3, 1)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, 1, 1, 1, c).permute(0, 2, 3, 1)
    xg = x.permute(0, 2, 3, 1).contiguous().float().cuda()
    dyg = dy.permute(0, 2, 3, 1).contiguous().float().cuda()
    wg = K.to_filter_layout(w.float().cuda())
    y = K.conv2d_fwd(xg, wg, None, 1, 1, c)
    dx = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, c)
    assert rel(y, y_ref) < 2e-6 and rel(dx, dx_ref) < 2e-6
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, 1, 1, 1, c)
    dw = K.conv2d_wgrad(xg, dyg, wg.shape, 1, 1, c)
    assert rel(dw, dw_ref) < 5e-6


@pytest.mark.parametrize("n,h,cexp,cout,seg,act", [(6, 28, 192, 48, 0, "silu"), (9, 14, 256, 64, 4, "silu"), (160, 28, 256, 64, 32, "silu"), (3, 7, 96, 24, 1, "relu")])
def test_pointwise_dgrad_reduces_batchnorm_backward_sums(K, n, h, cexp, cout, seg, act):
    """suhpe_conv2d_dgrad_bnbwd (FusedMBConv backward): the project conv's input-gradient GEMM returns g = dgrad(dt) * act'(BN(y0)) and the
    expand BatchNorm's backward sums from its epilogue; g, the sums (fp64 reference, one and two batch segments, a segment boundary inside a
    128-row tile) and the BatchNorm backward assembled from them (apply pass with activation 'none') equal the unfused chain
    dgrad -> three-pass BatchNorm backward; dgamma / dbeta are accumulated on top of what the buffers held"""
    g_ = torch.Generator().manual_seed(n * 7 + cexp)
    y0 = (torch.randn(n, h, h, cexp, generator=g_) * 1.4 + 0.2).cuda()              # expand conv output = the BatchNorm's input
    dt = torch.randn(n, h, h, cout, generator=g_).cuda()                             # gradient of the project conv's output
    w = K.to_filter_layout((torch.randn(cout, cexp, 1, 1, generator=g_) / 8).cuda())
    gamma, beta = (torch.rand(cexp, generator=g_) + 0.5).cuda(), (torch.randn(cexp, generator=g_) * 0.2).cuda()
    rows, hw = n * h * h, h * h
    if seg:
        y0[seg:] = y0[seg:] * 0.7 - 0.3
        mean, rstd = K.bn_stats_seg(y0, seg * hw, cexp, 1e-3, 0.1)
    else:
        mean, rstd = K.bn_stats(y0, cexp, 1e-3, 0.1)
    sc, sh = K.bn_affine_from_stats(mean, rstd, gamma, beta)
    dz_ref = K.conv2d_dgrad(dt, w, y0.shape, 1, 0, 1)                                # gradient of the activated tensor
    dg_ref, db_ref = torch.full((cexp,), 0.5, device="cuda"), torch.full((cexp,), -0.25, device="cuda")
    dg, db = dg_ref.clone(), db_ref.clone()
    if seg:
        dy_ref = K.bn_act_bwd_seg(y0, dz_ref, seg * hw, cexp, mean, rstd, gamma, beta, act, dg_ref, db_ref, True)
    else:
        dy_ref = K.bn_act_bwd(y0, dz_ref, cexp, mean, rstd, gamma, beta, act, True, dg_ref, db_ref, True)
    g, sums = K.conv2d_dgrad_bnbwd(dt, w, y0.shape, y0, mean, rstd, sc, sh, act, seg, dg, db)
    if seg:
        dy = K.bn_act_bwd_apply_seg(y0, g, seg * hw, cexp, mean, rstd, gamma, beta, "none", sums)
    else:
        dy = K.bn_act_bwd_apply(y0, g, cexp, mean, rstd, gamma, beta, "none", sums, rows)
    assert rel(dy, dy_ref.cpu()) < 5e-6
    assert rel(dg, dg_ref.cpu()) < 2e-5 and rel(db, db_ref.cpu()) < 2e-5
    xs = y0.double().cpu().reshape(rows, cexp)
    parts = [(0, rows)] if not seg else [(0, seg * hw), (seg * hw, rows)]
    m64, r64 = mean.double().cpu().reshape(-1, cexp), rstd.double().cpu().reshape(-1, cexp)
    g64 = torch.empty(rows, cexp, dtype=torch.float64)
    for i, (lo, hi) in enumerate(parts):
        xh = (xs[lo:hi] - m64[i]) * r64[i]
        u = xh * gamma.double().cpu() + beta.double().cpu()
        ga = (torch.sigmoid(u) * (1 + u * (1 - torch.sigmoid(u)))) if act == "silu" else (u > 0).double()
        g64[lo:hi] = dz_ref.double().cpu().reshape(rows, cexp)[lo:hi] * ga
        s = sums.reshape(-1, cexp, 2)[i].double().cpu()
        assert rel(s[:, 0], g64[lo:hi].sum(0)) < 2e-5 and rel(s[:, 1], (g64[lo:hi] * xh).sum(0)) < 2e-5
    assert rel(g.reshape(rows, cexp), g64) < 3e-6


def test_masks_of_a_pass_from_one_launch_equal_consecutive_draws(K):
    """suhpe_dropout_mask_batch / layers.prefetch_masks: the StochasticDepth masks of a forward pass drawn ahead by one launch are the values
    block-by-block draws give (same stream positions), the stream position after them is the same, a request of odd length included; a draw
    that does not match the head of the queue falls back to an ordinary draw"""
    from semiuhpe_amd import layers as L
    dev = torch.device("cuda", torch.cuda.current_device())
    reqs = [((160,), 0.995), ((160,), 0.9), ((7,), 0.5), ((160, 3), 0.8), ((160,), 0.85)]
    K.rng_counter(dev, seed=77)
    seq = [K.dropout_mask(s, k, dev).cpu() for s, k in reqs]
    after_seq = K.dropout_mask((64,), 0.5, dev).cpu()
    K.rng_counter(dev, seed=77)
    bat = [m.cpu() for m in K.dropout_masks(reqs, dev)]
    after_bat = K.dropout_mask((64,), 0.5, dev).cpu()
    for a, b in zip(seq, bat):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(after_seq, after_bat)
    K.rng_counter(dev, seed=77)
    L.prefetch_masks([("stochastic_depth", s, k) for s, k in reqs], dev)
    got = [L.draw_mask("stochastic_depth", s, k, dev).cpu() for s, k in reqs[:3]]
    for a, b in zip(seq, got):
        assert torch.equal(a, b)
    other = L.draw_mask("dropout", (5, 5), 0.5, dev)       # not the head of the queue: ordinary draw
    assert other.shape == (5, 5) and L._mask_queue is None


@pytest.mark.parametrize("n,hw,cin,cout", [(4, 56, 48, 192), (3, 24, 32, 96), (2, 16, 48, 40), (5, 8, 96, 48)])
def test_halo_tile_of_48_columns(K, n, hw, cin, cout):
    """igemm_nt3x3_n48_kernel (conv_halo.inl): 128 x 48 tiles whose columns 32-47 run on v_mfma_f32_16x16x4_f32 blocks.  Pinned for the forward
    (cout columns: one or two 48-column tiles, a partly filled 16-column block for cout = 40) and for the input gradient (cin columns, with and
    without the skip-connection add) of 3x3 / stride 1 / pad 1 convs on PATCH geometries; results against the heuristic tile (other summation
    order inside a slab: 3e-6) and against torch in fp64."""
    import ctypes
    import torch.nn.functional as F
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
    g = torch.Generator().manual_seed(n * 31 + cin)
    x = torch.randn(n, cin, hw, hw, generator=g, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / (3 * cin ** 0.5)
    dy = torch.randn(n, cout, hw, hw, generator=g, dtype=torch.float64)
    skip = torch.randn(n, cin, hw, hw, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, w, None, 1, 1).permute(0, 2, 3, 1)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, 1, 1).permute(0, 2, 3, 1)
    xg = x.permute(0, 2, 3, 1).contiguous().float().cuda()
    dyg = dy.permute(0, 2, 3, 1).contiguous().float().cuda()
    sg = skip.permute(0, 2, 3, 1).contiguous().float().cuda()
    wg = K.to_filter_layout(w.float().cuda())
    d = K.conv_desc(xg.shape, wg.shape, 1, 1, 1)
    names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
    pin = names.index("igemm_nt3x3_n48_kernel<128,48>")
    y0 = K.conv2d_fwd(xg, wg, None, 1, 1, 1)
    dx0 = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, 1)
    dxs0 = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, 1, skip=sg)
    try:
        pinned = 0
        for pid in (0, 1):                                   # (cout = 40 cannot be chunked for the input gradient's reduction: forward only there)
            if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, pin) == 0:
                assert lib.suhpe_conv2d_kernel_name(ctypes.byref(d), pid).decode() == "igemm_nt3x3_n48_kernel<128,48>"
                pinned += 1
        assert pinned == (1 if cout == 40 else 2)
        y = K.conv2d_fwd(xg, wg, None, 1, 1, 1)
        dx = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, 1)
        dxs = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, 1, skip=sg)
        # with statistics the tile falls back to the igemm tile of its record height: same bits as the 128x64 tile
        ys, part, tiles = K.conv2d_fwd_stats(xg, wg, 1, 1, 1)
    finally:
        for pid in (0, 1):
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    assert rel(y, y0.cpu()) < 3e-6 and rel(dx, dx0.cpu()) < 3e-6 and rel(dxs, dxs0.cpu()) < 3e-6
    assert rel(y, y_ref) < 1e-5 and rel(dx, dx_ref) < 1e-5 and rel(dxs, dx_ref + skip.permute(0, 2, 3, 1)) < 1e-5
    assert rel(ys, y_ref) < 1e-5 and tiles == (n * hw * hw + 127) // 128


@pytest.mark.parametrize("n,tokens,cin,cout,keep", [(12, 196, 384, 384, 0.8), (9, 49, 1536, 384, None), (5, 196, 576, 576, 0.9), (3, 64, 100, 36, 0.5)])
def test_linear_closes_residual_connection_in_epilogue(K, n, tokens, cin, cout, keep):
    """suhpe_conv2d_fwd_skip (TinyViT: x + drop_path(proj(.)), x + drop_path(fc2(.))): y = (x W^T + b) * keep_scale[sample] + skip from the GEMM
    epilogue equals Linear followed by the add pass (kernels.add_rowscale), with and without a keep scale, on the heuristic tile and with
    persistent / resident / wide tiles pinned (remapped to the igemm tile that carries the epilogue when a scale is present)"""
    import ctypes
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n * tokens, 1, 1, cin, generator=g).cuda()
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    skip = torch.randn(n, tokens, cout, generator=g).cuda()
    rs = None if keep is None else ((torch.rand(n, generator=g) < keep).float() / keep).cuda()
    y_ref = K.add_rowscale(skip, K.conv2d_fwd(x, w, b, 1, 0, 1).view(n, tokens, cout), rs)
    d = K.conv_desc(x.shape, w.shape, 1, 0, 1)
    names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
    try:
        for pin in (None, "gemm_nt_persistent<64,64,32,32>", "gemm_nt_resident<64,2,8>", "igemm_nt_kernel<128,128,64,64,4>", "igemm_nt_kernel<128,32,32,32,4>"):
            if pin is not None and lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names.index(pin)) != 0:
                continue
            y = K.conv2d_fwd_skip(x, w, b, skip, rs, tokens)
            assert rel(y.view(n, tokens, cout), y_ref.cpu()) < 1e-6, pin
    finally:
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
    yd = (x.double().cpu().view(-1, cin) @ w.double().cpu().view(cout, cin).t() + b.double().cpu()).view(n, tokens, cout)
    if rs is not None:
        yd = yd * rs.double().cpu()[:, None, None]
    assert rel(y.view(n, tokens, cout), yd + skip.double().cpu()) < 2e-5


@pytest.mark.parametrize("c,tiles,first,boundary", [(24, 4000, 4000, None), (24, 4000, 801, True), (40, 37, 12, True), (192, 3100, 620, False),
                                                   (8, 65, 65, True), (12, 1, 1, True), (16, 2500, 2500, None)], ids=str)
def test_bn_statistics_from_long_record_lists(K, c, tiles, first, boundary):
    """suhpe_bn_stats_from_partials_consume (runs of records folded all over the chip, then <= 32 fp64 pairs per channel and segment) gives the
    statistics and running-stat updates of the one-launch finalizers and of float64, for one and two segments, with and without the boundary
    record, for run lengths that do and do not divide the record counts.  boundary None = the one-segment form."""
    from semiuhpe_amd import kernels as KK
    g = torch.Generator().manual_seed(c * 1000 + tiles)
    rows_per = 16
    x = torch.randn(tiles + 1, rows_per, c, generator=g) * (0.5 + torch.rand(c, generator=g)) + 3.0 * torch.randn(c, generator=g)
    k = x[:, 0, :]
    d = x - k[:, None, :]
    rec = torch.stack([d.sum(1), (d * d).sum(1), k, torch.full_like(k, float(rows_per))], dim=-1).contiguous()       # [tile][c][4]
    part = rec[:tiles].cuda()
    extra = rec[tiles].contiguous().cuda() if boundary else None
    two = boundary is not None
    rows_a = first * rows_per if two else tiles * rows_per
    rows = tiles * rows_per + (rows_per if boundary else 0)
    xa = x[:first].reshape(-1, c).double() if two else x[:tiles].reshape(-1, c).double()
    xb = torch.cat([x[first:tiles].reshape(-1, c), x[tiles].reshape(-1, c) if boundary else x[:0].reshape(-1, c)]).double() if two else None
    eps, mom = 1e-3, 0.1

    def fresh():
        return (torch.full((c,), 0.25, device="cuda"), torch.full((c,), 1.5, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda"))

    def run(fold_min):
        old, KK._FOLD_MIN_RECORDS = KK._FOLD_MIN_RECORDS, fold_min
        try:
            rm, rv, nb = fresh()
            p = part.clone()
            if two:
                m, r = K.bn_stats_from_partials_seg(p, tiles, first, extra, rows, rows_a, c, eps, mom, rm, rv, nb)
            else:
                m, r = K.bn_stats_from_partials(p, tiles, rows, c, eps, mom, rm, rv, nb)
            torch.cuda.synchronize()
            return m.cpu(), r.cpu(), rm.cpu(), rv.cpu(), int(nb), p
        finally:
            KK._FOLD_MIN_RECORDS = old
    plain, folded = run(1 << 30), run(1)
    assert torch.equal(plain[5], part) and (tiles < 2 or not torch.equal(folded[5], part))          # only the folding form consumes its records
    for a, b in zip(plain[:4], folded[:4]):
        assert rel(b, a) < 1e-6
    assert plain[4] == folded[4] == (2 if two else 1)
    segs = [xa, xb] if two else [xa]
    for i, xs in enumerate(segs):
        m = folded[0][i] if two else folded[0]
        r = folded[1][i] if two else folded[1]
        assert rel(m, xs.mean(0)) < 1e-5 and rel(r, (xs.var(0, unbiased=False) + eps).rsqrt()) < 2e-5
