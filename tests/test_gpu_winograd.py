"""Winograd F(2x2, 3x3) / F(3x3, 2x2) kernels (csrc/conv_wino.hip) behind the 3x3 / stride 1 convolutions of torchvision's FusedMBConv
(reference src/networks.py:303) and of RepVGGBlock.rbr_dense (src/repvgg.py:28-32,67-76):

  * kernel level: forward (+ bias), input gradient (+ skip add), filter gradient (+ accumulate) and the BatchNorm statistics epilogue (one and
    two batch segments, the boundary inside a 64-tile block) against float64 torch references, on shapes that exercise ragged tile blocks,
    groups, partial channel chunks and output-channel padding.  Tolerance 2e-5 of the tensor's max (the suite's fp32 bar; measured ~3e-7:
    the Winograd sums are CLOSER to float64 than the direct kernels', fewer accumulation steps per output);
  * network level: the parity suite runs on the heuristic tiles, which never pick a Winograd tile; the bench runs on the tile database, which
    pins them for most 3x3 shapes.  SUHPE_WINOGRAD=2 makes the heuristics take the Winograd tiles wherever the geometry allows, and the
    network-level parity tests of the two backbones that have such convs are re-run that way in a child process.
"""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K():
    from semiuhpe_amd import kernels
    return kernels


def rel(x, ref):
    x = x.detach().cpu().double()
    ref = ref.detach().cpu().double()
    return float((x - ref).abs().max() / max(1.0, float(ref.abs().max())))


def _pin(K, d, names_by_pass, wino=True):
    from semiuhpe_amd import _lib
    lib = _lib.load()
    for pid, name in ((0, "wino_nt_kernel<64,64>"), (1, "wino_nt_kernel<64,64>"), (2, "wino_wgrad_kernel<64,64>")):
        rc = lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, names_by_pass[pid].index(name) if wino else -1)
        assert rc == 0, (pid, rc)


def _names():
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    return {pid: [lib.suhpe_conv2d_tile_name(pid, i).decode() for i in range(lib.suhpe_conv2d_tile_count(pid))] for pid in (0, 1, 2)}


# n, cin, cout, groups, h, w, samples of the first batch segment
SHAPES = [
    (3, 8, 12, 1, 6, 6, 1),          # one ragged tile block, one chunk pair, 12 of 64 output channels
    (5, 20, 36, 1, 10, 10, 2),       # partial super-chunk (20 of 32 channels), segment boundary inside a block
    (7, 16, 16, 2, 4, 4, 3),         # groups: 8 channels per group
    (4, 48, 192, 1, 28, 28, 1),      # EfficientNetV2-S stage 2 (48 input channels: the 3-sub-block filter-gradient form), three cout blocks
    (6, 64, 256, 1, 14, 14, 2),      # stage 3: 294 tiles = 4.6 blocks, boundary at tile 98
    (2, 24, 24, 1, 16, 16, 1),       # stage 1 widths
    (3, 128, 128, 2, 12, 10, 1),     # RepVGG grouped block, non-square image
    (2, 256, 128, 1, 8, 8, 1),       # long reduction (16 super-chunks), two items per workgroup stream
    (8, 256, 256, 1, 10, 10, 3),     # ResNet-50 layer3 conv2 at 160-pixel inputs: 200 tiles, four filter-gradient slabs, 4 x 4 channel blocks
    (16, 16, 16, 1, 32, 32, 5),      # many tiles, one channel block: 64 split-K slabs of the filter gradient (the sixteen-wavefront finish kernel)
    (2, 16, 64, 1, 2, 18, 1),        # one row of tiles per image (odd heights / widths -- ResNet-50's layer4 at 7 x 7 -- are refused: tests/test_cabi.py)
    # the register-resident form (conv_wino_small.hip: 16 / 24 input channels, at most 32 outputs; the two 16 / 24-channel shapes above take it too)
    (5, 24, 24, 1, 10, 10, 2),       # 125 tiles = 7.8 blocks of 16, segment boundary at tile 50 = inside block 3 (both records of that block)
    (3, 48, 48, 2, 8, 8, 1),         # groups of 24 channels
    (4, 16, 20, 1, 6, 6, 1),         # 20 output channels: a partial second channel block
    (4, 24, 32, 1, 12, 12, 3),       # 32 output channels, boundary at tile 108 = inside block 6
    (2, 16, 8, 1, 14, 14, 1),        # 8 output channels: half of the epilogue lanes store nothing
    (3, 24, 24, 1, 4, 36, 1),        # two rows of 18 tiles per image: neighbour tiles across a 16-tile block and across the image border
]


@pytest.mark.parametrize("cfg", SHAPES, ids=str)
def test_winograd_kernels_match_float64(K, cfg):
    n, cin, cout, g, h, w, seg = cfg
    gen = torch.Generator().manual_seed(abs(hash(cfg)) % 1000)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin // g, 3, 3, generator=gen) / (3 * (cin // g) ** 0.5)
    bias = torch.randn(cout, generator=gen)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, None, 1, 1, 1, g)
    dy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dy.double())
    skip = torch.randn(n, h, w, cin, generator=gen)
    xg, dyg = x.permute(0, 2, 3, 1).contiguous().cuda(), dy.permute(0, 2, 3, 1).contiguous().cuda()
    wg = K.to_filter_layout(wt.cuda())
    d = K.conv_desc(xg.shape, wg.shape, 1, 1, g)
    names = _names()
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_kernel_name.restype = ctypes.c_char_p
    _pin(K, d, names, True)
    try:
        assert lib.suhpe_conv2d_wino_pinned(ctypes.byref(d), 0) == 1 and lib.suhpe_conv2d_wino_pinned(ctypes.byref(d), 1) == 1
        assert lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 2).decode() in ("wino_wgrad_kernel<64,64>", "wino_small_wgrad_kernel")      # (the second: at most 32 channels on both sides)
        nhwc = lambda t: t.permute(0, 2, 3, 1)
        y = K.conv2d_fwd(xg, wg, None, 1, 1, g)
        assert rel(y, nhwc(y_ref)) < 2e-5
        yb = K.conv2d_fwd(xg, wg, bias.cuda(), 1, 1, g)
        assert rel(yb, nhwc(y_ref) + bias.double()) < 2e-5
        dx = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, g)
        assert rel(dx, nhwc(xd.grad)) < 2e-5
        dxs = K.conv2d_dgrad(dyg, wg, xg.shape, 1, 1, g, skip=skip.cuda())
        assert rel(dxs, nhwc(xd.grad) + skip.double()) < 2e-5
        dw = K.conv2d_wgrad(xg, dyg, wg.shape, 1, 1, g)
        assert rel(dw, wd.grad) < 2e-5
        dw2 = K.conv2d_wgrad(xg, dyg, wg.shape, 1, 1, g, out=dw.clone(), accumulate=True)
        assert rel(dw2, 2 * wd.grad) < 2e-5
        # statistics epilogue: one segment, then two (same tensors as the plain forward, bit for bit)
        ys, part, tiles = K.conv2d_fwd_stats(xg, wg, 1, 1, g)
        assert torch.equal(ys, y)
        mean, rstd = K.bn_stats_from_partials(part, tiles, n * h * w, cout, 1e-3, 0.1)
        flat = y_ref.detach().permute(0, 2, 3, 1).reshape(-1, cout)
        assert rel(mean, flat.mean(0)) < 2e-5
        assert float(((rstd.cpu().double() - (flat.var(0, unbiased=False) + 1e-3).rsqrt()) * (flat.var(0, unbiased=False) + 1e-3).sqrt()).abs().max()) < 2e-5
        ys2, part2, tiles2, (first, extra) = K.conv2d_fwd_stats(xg, wg, 1, 1, g, seg=seg)
        assert torch.equal(ys2, y)
        m2, r2 = K.bn_stats_from_partials_seg(part2, tiles2, first, extra, n * h * w, seg * h * w, cout, 1e-3, 0.1)
        for s_, rows in ((0, y_ref.detach()[:seg]), (1, y_ref.detach()[seg:])):
            fl = rows.permute(0, 2, 3, 1).reshape(-1, cout)
            assert rel(m2[s_], fl.mean(0)) < 2e-5, s_
            assert float(((r2[s_].cpu().double() - (fl.var(0, unbiased=False) + 1e-3).rsqrt()) * (fl.var(0, unbiased=False) + 1e-3).sqrt()).abs().max()) < 2e-5, s_
    finally:
        _pin(K, d, names, False)


def test_winograd_tile_is_refused_where_it_cannot_run(K):
    """stride 2, 1x1, odd image sizes: suhpe_conv2d_set_tile refuses the Winograd tiles"""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    names = _names()
    for xs, ws, stride, pad in (((2, 8, 8, 16), (16, 16, 3, 3), 2, 1), ((2, 8, 8, 16), (16, 16, 1, 1), 1, 0), ((2, 7, 7, 16), (16, 16, 3, 3), 1, 1)):
        d = K.conv_desc(xs, ws, stride, pad, 1)
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names[0].index("wino_nt_kernel<64,64>")) != 0
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 2, names[2].index("wino_wgrad_kernel<64,64>")) != 0


@pytest.mark.parametrize("cfg", [(3, 16, 24, 1, 8, 8), (2, 32, 48, 1, 12, 10), (5, 48, 192, 1, 14, 14), (4, 128, 128, 2, 6, 6), (3, 64, 80, 1, 28, 28)], ids=str)
@pytest.mark.parametrize("act,res", [("silu", 0), ("relu", 1), ("none", 2), ("silu", 2), ("relu", 0)])
def test_winograd_eval_epilogue_matches_float64(K, cfg, act, res):
    """eval-mode conv -> folded BatchNorm -> activation (-> skip before / after it) on the Winograd kernel (suhpe_conv2d_fwd_affine_wino: the FusedMBConv
    blocks of an EfficientNetV2-S in eval(), RepVGG's re-parameterised blocks): against float64, and against the direct tile of the same call, for the
    one-block (<= 32 output channels), three-block (<= 48) and full forms of the kernel"""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    names = _names()
    n, cin, cout, g, h, w = cfg
    gen = torch.Generator().manual_seed(cin * 7 + cout + res)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin // g, 3, 3, generator=gen) / (3 * (cin // g) ** 0.5)
    scale, shift = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
    skip = torch.randn(n, cout, h, w, generator=gen)
    z = F.conv2d(x.double(), wt.double(), None, 1, 1, 1, g) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if res == 1:
        z = z + skip.double()
    z = {"silu": F.silu, "relu": F.relu, "none": lambda t: t}[act](z)
    if res == 2:
        z = z + skip.double()
    ref = z.permute(0, 2, 3, 1)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    wg = K.to_filter_layout(wt.cuda())
    sk = skip.permute(0, 2, 3, 1).contiguous().cuda() if res else None
    d = K.conv_desc(xg.shape, wg.shape, 1, 1, g)
    direct = K.conv2d_fwd_affine(xg, wg, scale.cuda(), shift.cuda(), act, sk, res == 2, stride=1, pad=1, groups=g)
    assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names[0].index("wino_nt_kernel<64,64>")) == 0
    try:
        assert lib.suhpe_conv2d_wino_pinned(ctypes.byref(d), 0) == 1
        out = K.conv2d_fwd_affine(xg, wg, scale.cuda(), shift.cuda(), act, sk, res == 2, stride=1, pad=1, groups=g)
    finally:
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
    assert rel(direct, ref) < 2e-5 and rel(out, ref) < 2e-5
    assert not torch.equal(out, direct)                     # (the Winograd kernel did run: another factorisation, other roundings)


def test_network_parity_suite_on_the_winograd_tiles():
    """second pass of the network-level parity tests with every eligible 3x3 conv on the Winograd kernels (forward, input gradient, filter gradient)"""
    # Two whole-net gradient yardsticks sit on ReLU-mask flips through 28-53 train-mode BatchNorm layers on 2-8 images -- chaotic quantities of which
    # ANY other rounding pattern draws a new sample (the Winograd kernels themselves are checked against float64 above at 3e-7).  Round 5 ran them
    # here with bars widened by half (2e-3 -> 3e-3, 6 x -> 9 x one fp32 sample's distance to float64); since round 6 they run with the SAME bars as on
    # the direct tiles, and the yardstick itself is sounder instead: the noise of the problem is the largest of THREE fp32 samples (the reference run
    # and two oracle runs on inputs moved by 1 ulp: parity_utils.fp32_noise_samples), not one.
    env = dict(os.environ, SUHPE_WINOGRAD="2", SUHPE_AUTOTUNE="0",
               SUHPE_BN_FOLD_MIN="6")      # (and every BatchNorm with six or more epilogue records on the folding finalizer the bench shapes take)
    tests = ["tests/test_gpu_effnet.py", "tests/test_gpu_agent_repvgg.py", "tests/test_gpu_resnet.py",
             "tests/test_gpu_backbones.py::test_repvgg_b1g2_vs_reference", "tests/test_gpu_backbones.py::test_repvgg_block_vs_reference",
             "tests/test_gpu_segments.py::test_segmented_conv_statistics_match_per_segment_sums",
             "tests/test_gpu_train_eval.py::test_train_then_eval"]       # train.py (both stages, validation, threshold scan, checkpoints) + eval.py
    # (test_gpu_segments' merged-pass-equals-two-passes tests compare two flows of this library through 28+ train-mode BatchNorm layers on 2 + 7
    # images of 64 pixels and sit within 20 % of their bars on the direct tiles already; the two-segment statistics of the Winograd epilogue are
    # checked above at kernel level.)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail


def test_winograd_results_do_not_depend_on_memory_latency(K):
    """The chunk loops of the Winograd kernels wait for their loads with hand-counted s_waitcnt vmcnt(N): a count that is one too high returns stale
    registers or reads an LDS image before its DMA has landed -- but only when the load is slow.  Here a second stream keeps HBM saturated (large copies)
    while forward, input gradient and filter gradient run repeatedly on bench-sized shapes: every result must be bit-identical to the quiet run."""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    names = _names()
    gen = torch.Generator().manual_seed(11)
    for n, cin, cout, hw in ((64, 48, 192, 56), (96, 64, 256, 28), (64, 512, 512, 14)):
        x = torch.randn(n, hw, hw, cin, generator=gen).cuda()
        dy = torch.randn(n, hw, hw, cout, generator=gen).cuda()
        wg = K.to_filter_layout((torch.randn(cout, cin, 3, 3, generator=gen) / (3 * cin ** 0.5)).cuda())
        d = K.conv_desc(x.shape, wg.shape, 1, 1, 1)
        _pin(K, d, names, True)
        try:
            quiet = (K.conv2d_fwd(x, wg, None, 1, 1, 1), K.conv2d_dgrad(dy, wg, x.shape, 1, 1, 1), K.conv2d_wgrad(x, dy, wg.shape, 1, 1, 1),
                     K.conv2d_fwd_stats(x, wg, 1, 1, 1)[1].clone())
            torch.cuda.synchronize()
            noise = torch.cuda.Stream()
            big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")           # 1 GiB: every copy below misses the caches
            with torch.cuda.stream(noise):
                for _ in range(60):
                    big[: 1 << 27].copy_(big[1 << 27:], non_blocking=True)
            for it in range(12):
                assert torch.equal(K.conv2d_fwd(x, wg, None, 1, 1, 1), quiet[0]), ("fwd", n, cin, it)
                assert torch.equal(K.conv2d_dgrad(dy, wg, x.shape, 1, 1, 1), quiet[1]), ("dgrad", n, cin, it)
                assert torch.equal(K.conv2d_wgrad(x, dy, wg.shape, 1, 1, 1), quiet[2]), ("wgrad", n, cin, it)
                assert torch.equal(K.conv2d_fwd_stats(x, wg, 1, 1, 1)[1], quiet[3]), ("stats records", n, cin, it)
            torch.cuda.synchronize()
            del big
        finally:
            _pin(K, d, names, False)


def test_last_kernel_name_reports_what_really_ran(K):
    """suhpe_conv2d_kernel_name predicts the kernel from the descriptor alone; run-time facts send a call elsewhere: a bias on a shape of the
    register-resident Winograd form runs wino_nt_kernel's narrow form instead (launch_wino_small declines a bias).  suhpe_last_kernel_name -- what
    kernels._Prof records, bench.py's per-kernel roofline and tools/pmc_traffic's name join -- must say so."""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    names = _names()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(6, 16, 16, 24, generator=gen).cuda()
    wg = K.to_filter_layout((torch.randn(24, 24, 3, 3, generator=gen) / 15).cuda())
    bias = torch.randn(24, generator=gen).cuda()
    d = K.conv_desc(x.shape, wg.shape, 1, 1, 1)
    _pin(K, d, names, True)
    try:
        predicted = lib.suhpe_conv2d_kernel_name(ctypes.byref(d), 0).decode()
        assert predicted == "wino_small_kernel"
        y0 = K.conv2d_fwd(x, wg, None, 1, 1, 1)
        assert lib.suhpe_last_kernel_name().decode() == "wino_small_kernel"
        y1 = K.conv2d_fwd(x, wg, bias, 1, 1, 1)
        assert lib.suhpe_last_kernel_name().decode() == "wino_nt_kernel<64,64>"          # the fallback the descriptor cannot see
        assert rel(y1 - bias.view(1, 1, 1, -1), y0) < 2e-6
        K.PROFILE = []
        try:
            K.conv2d_fwd(x, wg, bias, 1, 1, 1)
            assert [r[0] for r in K.PROFILE] == ["wino_nt_kernel<64,64>"]                # ... and the live profile keys on it
        finally:
            K.PROFILE = None
    finally:
        _pin(K, d, names, False)
    K.conv2d_fwd(x, wg, None, 1, 1, 1)
    assert lib.suhpe_last_kernel_name().decode().startswith(("igemm_nt", "gemm_nt"))     # unpinned: a direct tile
