"""Out-of-bounds writes of the round-4 kernels, through the C ABI with caller-owned buffers: every output (and every scratch buffer) sits between two guard
bands filled with a sentinel; after the launch the bands must be untouched and the payload fully written where the contract says so.  Ragged geometry is
where such bugs live: tile counts that are no multiple of a 64-tile item (or of a wavefront's 16-tile block), channel counts that are no multiple of a
64- / 32- / 16-channel block, groups, a batch segment that cuts a block.  (PyTorch's allocator rounds sizes up and packs tensors: a stray store usually
lands in a neighbour and no parity test of THIS op sees it.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
GUARD = 4096                  # floats on each side
SENT = -7.25e33


class Guarded:
    def __init__(self, floats):
        self.n = int(floats)
        self.buf = torch.full((self.n + 2 * GUARD,), SENT, dtype=torch.float32, device="cuda")

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + 4 * GUARD)

    @property
    def payload(self):
        return self.buf[GUARD:GUARD + self.n]

    def check(self, what, fully_written=True):
        torch.cuda.synchronize()
        assert bool((self.buf[:GUARD] == SENT).all()), f"{what}: store below the buffer"
        assert bool((self.buf[GUARD + self.n:] == SENT).all()), f"{what}: store beyond the buffer"
        if fully_written:
            assert not bool((self.payload == SENT).any()), f"{what}: part of the output was never written"


def _lib_and_names():
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    lib.suhpe_conv2d_wino_filter_bytes.restype = ctypes.c_int64
    lib.suhpe_conv2d_wgrad_workspace_bytes.restype = ctypes.c_int64
    names = {p: [lib.suhpe_conv2d_tile_name(p, i).decode() for i in range(lib.suhpe_conv2d_tile_count(p))] for p in (0, 1, 2)}
    return _lib, lib, names


# n, cin, cout, groups, h, w, images of the first batch segment
CASES = [(3, 8, 12, 1, 6, 6, 1), (5, 20, 36, 1, 10, 10, 2), (7, 16, 16, 2, 4, 4, 3), (2, 24, 24, 1, 18, 14, 1), (3, 48, 72, 1, 14, 22, 1),
         (5, 64, 200, 1, 6, 10, 2), (2, 136, 44, 1, 8, 8, 1), (9, 32, 32, 4, 2, 2, 4),
         # the register-resident form (conv_wino_small.hip): ragged last block, segment boundary inside a block, 20 / 8 output channels, groups of 24
         (5, 24, 24, 1, 10, 10, 2), (3, 16, 20, 1, 6, 6, 1), (3, 48, 48, 2, 8, 8, 1), (2, 16, 8, 1, 14, 14, 1), (3, 24, 32, 1, 4, 36, 2)]


@pytest.mark.parametrize("cfg", CASES, ids=str)
def test_winograd_entry_points_stay_inside_their_buffers(cfg):
    _lib, lib, names = _lib_and_names()
    from semiuhpe_amd import kernels as K
    n, cin, cout, g, h, w, seg = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(n, h, w, cin, generator=gen).cuda()
    dy = torch.randn(n, h, w, cout, generator=gen).cuda()
    wt = K.to_filter_layout((torch.randn(cout, cin // g, 3, 3, generator=gen) * 0.1).cuda())
    d = K.conv_desc(x.shape, wt.shape, 1, 1, g)
    dref = ctypes.byref(d)
    for pid, nm in ((0, "wino_nt_kernel<64,64>"), (1, "wino_nt_kernel<64,64>"), (2, "wino_wgrad_kernel<64,64>")):
        assert lib.suhpe_conv2d_set_tile(dref, pid, names[pid].index(nm)) == 0
    try:
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        # transformed filters (forward / input gradient)
        fw = [Guarded(lib.suhpe_conv2d_wino_filter_bytes(dref, p) // 4) for p in (0, 1)]
        for p in (0, 1):
            _lib.call("suhpe_conv2d_wino_filter", dref, p, P(wt), fw[p].ptr)
            fw[p].check(f"wino_filter pass {p}")
        # forward + statistics records of two segments (the boundary record only exists when the segment cuts a 16-tile block)
        tiles = lib.suhpe_conv2d_fwd_stats_tiles(dref)
        y, st, ex = Guarded(n * h * w * cout), Guarded(cout * tiles * 4), Guarded(cout * 4)
        _lib.call("suhpe_conv2d_fwd_wino", dref, P(x), P(wt), fw[0].ptr, None, y.ptr, st.ptr, ex.ptr, ctypes.c_int64(seg * h * w))
        y.check("fwd_wino y")
        st.check("fwd_wino stats")
        ex.check("fwd_wino boundary record", fully_written=False)
        # eval-mode epilogue
        sc, sh = torch.rand(cout, generator=gen).cuda() + 0.5, torch.randn(cout, generator=gen).cuda()
        ya = Guarded(n * h * w * cout)
        _lib.call("suhpe_conv2d_fwd_affine_wino", dref, P(x), P(wt), fw[0].ptr, P(sc), P(sh), 3, y.ptr, 1, ya.ptr)
        ya.check("fwd_affine_wino y")
        # input gradient + skip (no transposed filter)
        dx = Guarded(n * h * w * cin)
        _lib.call("suhpe_conv2d_dgrad_wino", dref, P(dy), None, fw[1].ptr, P(x), dx.ptr)
        dx.check("dgrad_wino dx")
        # filter gradient: split-K slabs + finish kernel in the caller's workspace, then the reduction into dw
        wsb = lib.suhpe_conv2d_wgrad_workspace_bytes(dref)
        ws, dw = Guarded(wsb // 4), Guarded(wt.numel())
        _lib.call("suhpe_conv2d_wgrad", dref, P(x), P(dy), dw.ptr, 0, ws.ptr, ctypes.c_int64(wsb))
        ws.check("wgrad workspace", fully_written=False)
        dw.check("wgrad dw")
        # results against the direct tiles (the guard bands did not change what the kernels compute)
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_set_tile(dref, pid, -1)
        ref_y = K.conv2d_fwd(x, wt, None, 1, 1, g)
        ref_dx = K.conv2d_dgrad(dy, wt, x.shape, 1, 1, g, skip=x)
        ref_dw = K.conv2d_wgrad(x, dy, wt.shape, 1, 1, g)
        rel = lambda a, b: float((a.double() - b.double().reshape(a.shape)).abs().max() / max(1.0, float(b.abs().max())))
        assert rel(y.payload, ref_y) < 2e-5 and rel(dx.payload, ref_dx) < 2e-5
        assert rel(dw.payload, ref_dw.permute(0, 2, 3, 1)) < 2e-5          # (filters are stored [cout][r][s][cin / groups])
    finally:
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_set_tile(dref, pid, -1)


@pytest.mark.parametrize("c,tiles,first,boundary", [(24, 4000, 801, True), (12, 65, 65, True), (40, 37, 12, False), (192, 3100, 3100, None)], ids=str)
def test_consuming_finalizer_stays_inside_its_buffers(c, tiles, first, boundary):
    _lib, lib, _ = _lib_and_names()
    gen = torch.Generator().manual_seed(c + tiles)
    rec = Guarded(tiles * c * 4)
    rec.payload.copy_(torch.randn(tiles * c * 4, generator=gen).cuda())
    rec.payload.view(tiles, c, 4)[:, :, 3] = 16.0
    two = boundary is not None
    extra = torch.randn(c, 4, generator=gen).cuda() if boundary else None
    if extra is not None:
        extra[:, 3] = 16.0
    rows = 16 * (tiles + (1 if boundary else 0))
    rows_a = 16 * first if two else rows
    mean, rstd, rm, rv = Guarded((2 if two else 1) * c), Guarded((2 if two else 1) * c), Guarded(c), Guarded(c)
    rm.payload.fill_(0.0)
    rv.payload.fill_(1.0)
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    _lib.call("suhpe_bn_stats_from_partials_consume", rec.ptr, tiles, first if two else tiles, ctypes.c_void_p(extra.data_ptr()) if boundary else None,
              ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, ctypes.c_float(1e-3), ctypes.c_float(0.1), mean.ptr, rstd.ptr, rm.ptr, rv.ptr,
              ctypes.c_void_p(nbt.data_ptr()))
    for gbuf, what in ((rec, "records"), (mean, "mean"), (rstd, "rstd"), (rm, "running_mean"), (rv, "running_var")):
        gbuf.check(what)
    assert int(nbt) == (2 if two else 1)


# n, cin, cout, k, stride, pad, groups, h, w: row counts off every tile height, channel counts off every tile width, strides, groups
MENU_CASES = [(3, 20, 36, 1, 1, 0, 1, 9, 7), (2, 40, 24, 3, 1, 1, 1, 9, 11), (5, 16, 72, 3, 2, 1, 1, 13, 9), (2, 64, 48, 3, 1, 1, 2, 6, 10),
              (37, 256, 132, 1, 1, 0, 1, 3, 3), (1, 8, 520, 1, 1, 0, 1, 5, 5)]


@pytest.mark.parametrize("cfg", MENU_CASES, ids=str)
def test_every_gemm_tile_stays_inside_its_buffers(cfg):
    """every entry of the three tile menus that accepts the geometry (direct, halo, 48-column, persistent, resident, Winograd), forward / input gradient /
    filter gradient, outputs and the split-K workspace between guard bands"""
    _lib, lib, names = _lib_and_names()
    from semiuhpe_amd import kernels as K
    n, cin, cout, k, stride, pad, g, h, w = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(n, h, w, cin, generator=gen).cuda()
    wt = K.to_filter_layout((torch.randn(cout, cin // g, k, k, generator=gen) * 0.1).cuda())
    d = K.conv_desc(x.shape, wt.shape, stride, pad, g)
    dref = ctypes.byref(d)
    dy = torch.randn(n, d.ho, d.wo, cout, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ref_y = K.conv2d_fwd(x, wt, None, stride, pad, g)
    ref_dx = K.conv2d_dgrad(dy, wt, x.shape, stride, pad, g)
    ref_dw = K.conv2d_wgrad(x, dy, wt.shape, stride, pad, g)
    wT = torch.empty_like(wt)
    _lib.call("suhpe_conv2d_transpose_filter", dref, P(wt), P(wT))
    rel = lambda a, b: float((a.double() - b.double().reshape(a.shape)).abs().max() / max(1.0, float(b.abs().max())))
    ran = 0
    try:
        for pid in (0, 1, 2):
            for idx, name in enumerate(names[pid]):
                if lib.suhpe_conv2d_set_tile(dref, pid, idx) != 0:
                    continue                                    # this tile does not take the geometry
                ran += 1
                if pid == 0:
                    y = Guarded(ref_y.numel())
                    _lib.call("suhpe_conv2d_fwd", dref, P(x), P(wt), None, y.ptr)
                    y.check(f"fwd {name}")
                    assert rel(y.payload, ref_y) < 2e-5, name
                elif pid == 1:
                    dx = Guarded(x.numel())
                    _lib.call("suhpe_conv2d_dgrad", dref, P(dy), P(wT), dx.ptr)
                    dx.check(f"dgrad {name}")
                    assert rel(dx.payload, ref_dx) < 2e-5, name
                else:
                    wsb = lib.suhpe_conv2d_wgrad_workspace_bytes(dref)
                    ws, dw = Guarded(max(1, wsb // 4)), Guarded(wt.numel())
                    _lib.call("suhpe_conv2d_wgrad", dref, P(x), P(dy), dw.ptr, 0, ws.ptr, ctypes.c_int64(wsb))
                    ws.check(f"wgrad workspace {name}", fully_written=False)
                    dw.check(f"wgrad {name}")
                    assert rel(dw.payload, ref_dw.permute(0, 2, 3, 1)) < 2e-5, name
                lib.suhpe_conv2d_set_tile(dref, pid, -1)
    finally:
        for pid in (0, 1, 2):
            lib.suhpe_conv2d_set_tile(dref, pid, -1)
    assert ran >= 12


# n, c, h, w, stride: the image-tile kernels (14 x 14, 7 x 7), the sliding-window kernels, odd sizes, stride 2, few channels
DW_CASES = [(5, 24, 14, 14, 1), (9, 40, 7, 7, 1), (3, 16, 28, 28, 1), (2, 8, 56, 56, 1), (3, 12, 15, 9, 1), (4, 20, 14, 14, 2), (2, 36, 13, 13, 2), (33, 4, 7, 7, 1)]


@pytest.mark.parametrize("cfg", DW_CASES, ids=str)
def test_depthwise_kernels_stay_inside_their_buffers(cfg):
    """depthwise 3x3: forward, forward + statistics records (two segments), forward with the producer's BatchNorm + SiLU in the loader, input gradient,
    filter gradient (+ its workspace)"""
    _lib, lib, _ = _lib_and_names()
    from semiuhpe_amd import kernels as K
    lib.suhpe_dwconv_wgrad_workspace_bytes.restype = ctypes.c_int64
    n, c, h, w, stride = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(n, h, w, c, generator=gen).cuda()
    wt = (torch.randn(c, 1, 3, 3, generator=gen) * 0.3).cuda()
    d = K.conv_desc(x.shape, wt.shape, stride, 1, c)
    dref = ctypes.byref(d)
    dy = torch.randn(n, d.ho, d.wo, c, generator=gen).cuda()
    mean, rstd = torch.randn(2, c, generator=gen).cuda(), (torch.rand(2, c, generator=gen) + 0.5).cuda()
    gamma, beta = torch.randn(c, generator=gen).cuda(), torch.randn(c, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ysz = n * d.ho * d.wo * c
    y = Guarded(ysz)
    _lib.call("suhpe_dwconv_fwd", dref, P(x), P(wt), y.ptr)
    y.check("dwconv_fwd")
    blocks = lib.suhpe_dwconv_fwd_stats_blocks(dref)
    seg = max(1, n // 3)
    if blocks > 0:
        y2, st = Guarded(ysz), Guarded(2 * blocks * c * 4)
        _lib.call("suhpe_dwconv_fwd_stats", dref, P(x), P(wt), y2.ptr, st.ptr, seg)
        y2.check("dwconv_fwd_stats y")
        st.check("dwconv_fwd_stats records")
        assert float((y2.payload - y.payload).abs().max()) < 1e-5          # (another kernel instance: other summation order)
        y3, st3 = Guarded(ysz), Guarded(2 * blocks * c * 4)
        _lib.call("suhpe_dwconv_bnact_fwd", dref, P(x), P(mean), P(rstd), P(gamma), P(beta), 3, seg, P(wt), y3.ptr, st3.ptr)
        y3.check("dwconv_bnact_fwd y")
        st3.check("dwconv_bnact_fwd records")
    dx = Guarded(x.numel())
    _lib.call("suhpe_dwconv_dgrad", dref, P(dy), P(wt), dx.ptr)
    dx.check("dwconv_dgrad")
    wsb = lib.suhpe_dwconv_wgrad_workspace_bytes(dref)
    ws, dw = Guarded(max(1, wsb // 4)), Guarded(wt.numel())
    _lib.call("suhpe_dwconv_wgrad", dref, P(x), P(dy), dw.ptr, 0, ws.ptr, ctypes.c_int64(wsb))
    ws.check("dwconv_wgrad workspace", fully_written=False)
    dw.check("dwconv_wgrad")
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), None, stride, 1, 1, c).permute(0, 2, 3, 1)
    assert float((y.payload.cpu().double() - ref.reshape(-1)).abs().max()) < 1e-4
    # the whole backward in one launch (round 6): plain form and section form (the output gradient formed from (y, dout, s, dpool, chan) on the fly)
    lib.suhpe_dwconv_bwd_fused_workspace_bytes.restype = ctypes.c_int64
    if lib.suhpe_dwconv_bwd_fused_ok(dref):
        fb = lib.suhpe_dwconv_bwd_fused_workspace_bytes(dref)
        s_ = torch.rand(n, c, generator=gen).cuda()
        dpool = torch.randn(n, c, generator=gen).cuda()
        chan = torch.randn(2, 2 * c, generator=gen).cuda()
        for section in (False, True):
            dx2, sums, dgm, dbt, dw2, ws2 = Guarded(x.numel()), Guarded(2 * 2 * c), Guarded(c), Guarded(c), Guarded(wt.numel()), Guarded(max(1, fb // 4))
            for t_ in (dgm, dbt, dw2):
                t_.payload.zero_()
            mid = (P(x), P(s_), P(dpool), P(chan), P(mean), P(rstd), P(gamma), P(beta), 3, 1) if section else (None,) * 8 + (0, 0)
            _lib.call("suhpe_dwconv_bwd_fused", dref, P(dy), P(wt), dx2.ptr, P(x), P(mean), P(rstd), P(gamma), P(beta), 3, seg, sums.ptr, dgm.ptr, dbt.ptr, 1,
                      dw2.ptr, 1, *mid, ws2.ptr, ctypes.c_int64(fb))
            dx2.check(f"dwconv_bwd_fused dx section={section}"); sums.check("dwconv_bwd_fused sums"); dgm.check("dwconv_bwd_fused dgamma")
            dbt.check("dwconv_bwd_fused dbeta"); dw2.check("dwconv_bwd_fused dw"); ws2.check("dwconv_bwd_fused workspace", fully_written=False)
            if not section:
                assert float((dx2.payload - dx.payload).abs().max()) < 1e-5


# rows, c, hw (rows per sample): channel counts off the 4-wide vectors and the 64-column blocks, row counts off every block height
BN_CASES = [(37 * 9, 24, 9), (5 * 49, 10, 49), (3 * 196, 132, 196), (1000, 7, 1), (64 * 16, 520, 16), (2 * 3136, 48, 3136)]


@pytest.mark.parametrize("cfg", BN_CASES, ids=str)
def test_batchnorm_kernels_stay_inside_their_buffers(cfg):
    """train-mode BatchNorm: statistics (one and two segments) with their workspace, normalise + SiLU (+ skip), backward with its workspace"""
    _lib, lib, _ = _lib_and_names()
    lib.suhpe_bn_workspace_bytes.restype = ctypes.c_int64
    rows, c, hw = cfg
    gen = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=gen).cuda()
    dy = torch.randn(rows, c, generator=gen).cuda()
    gamma, beta = torch.randn(c, generator=gen).cuda(), torch.randn(c, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    wsf = lib.suhpe_bn_workspace_bytes(c) // 4
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    ws, mean, rstd, rm, rv = Guarded(wsf), Guarded(c), Guarded(c), Guarded(c), Guarded(c)
    rm.payload.zero_(); rv.payload.fill_(1.0)
    _lib.call("suhpe_bn_stats", P(x), ctypes.c_int64(rows), c, ctypes.c_float(1e-3), ctypes.c_float(0.1), mean.ptr, rstd.ptr, rm.ptr, rv.ptr, P(nbt), ws.ptr)
    ws.check("bn_stats workspace", fully_written=False)
    for gb, what in ((mean, "mean"), (rstd, "rstd"), (rm, "running_mean"), (rv, "running_var")):
        gb.check("bn_stats " + what)
    assert float((mean.payload.cpu().double() - x.double().cpu().mean(0)).abs().max()) < 1e-5
    y = Guarded(rows * c)
    _lib.call("suhpe_bn_act_fwd", P(x), ctypes.c_int64(rows), c, mean.ptr, rstd.ptr, P(gamma), P(beta), 3, P(dy), y.ptr)
    y.check("bn_act_fwd")
    ws2, dx, dg, db = Guarded(wsf), Guarded(rows * c), Guarded(c), Guarded(c)
    _lib.call("suhpe_bn_act_bwd", P(x), P(dy), None, ctypes.c_int64(rows), c, mean.ptr, rstd.ptr, P(gamma), P(beta), 3, 1, dx.ptr, dg.ptr, db.ptr, 0, ws2.ptr)
    ws2.check("bn_act_bwd workspace", fully_written=False)
    for gb, what in ((dx, "dx"), (dg, "dgamma"), (db, "dbeta")):
        gb.check("bn_act_bwd " + what)
    if rows // hw >= 2:                                             # two batch segments (whole samples each)
        rows_a = (rows // hw // 2) * hw
        ws3, m2, r2 = Guarded(wsf), Guarded(2 * c), Guarded(2 * c)
        _lib.call("suhpe_bn_stats_seg", P(x), ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, ctypes.c_float(1e-3), ctypes.c_float(0.1), m2.ptr, r2.ptr,
                  rm.ptr, rv.ptr, P(nbt), ws3.ptr)
        ws3.check("bn_stats_seg workspace", fully_written=False)
        m2.check("bn_stats_seg mean"); r2.check("bn_stats_seg rstd")
        y2 = Guarded(rows * c)
        _lib.call("suhpe_bn_act_fwd_seg", P(x), ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, m2.ptr, r2.ptr, P(gamma), P(beta), 3, None, 0, None, hw, y2.ptr)
        y2.check("bn_act_fwd_seg")


# B, H, W, window, heads: TinyViT's windows (7 x 7 on 7 / 14 / 28-wide maps, 14 x 14 on the 14-wide map)
ATTN_CASES = [(3, 14, 14, 7, 2), (2, 28, 28, 7, 3), (5, 7, 7, 7, 1), (2, 14, 14, 14, 4), (1, 28, 14, 7, 18)]


@pytest.mark.parametrize("cfg", ATTN_CASES, ids=str)
def test_window_attention_stays_inside_its_buffers(cfg):
    """windowed attention forward (output, saved probabilities, bias workspace) and backward (dqkv, score-gradient workspace, bias-table gradient)"""
    _lib, lib, _ = _lib_and_names()
    from semiuhpe_amd.src.tiny_vit import attention_bias_index
    for f in ("suhpe_window_attn_fwd_workspace_bytes", "suhpe_window_attn_probs_elems", "suhpe_window_attn_dbias_workspace_bytes"):
        getattr(lib, f).restype = ctypes.c_int64
    B, H, W, ws, heads = cfg
    dh, N = 32, ws * ws
    gen = torch.Generator().manual_seed(sum(cfg))
    idx, noff = attention_bias_index(ws)
    bias_idx = idx.to(torch.int32).cuda().contiguous()
    qkv = torch.randn(B * H * W, heads * 3 * dh, generator=gen).cuda()
    table = torch.randn(heads, noff, generator=gen).cuda()
    dout = torch.randn(B * H * W, heads * dh, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    pe = lib.suhpe_window_attn_probs_elems(B, H, W, ws, heads)
    out, probs, wsf = Guarded(B * H * W * heads * dh), Guarded(pe), Guarded(lib.suhpe_window_attn_fwd_workspace_bytes(ws, heads) // 4)
    _lib.call("suhpe_window_attn_fwd", P(qkv), P(table), P(bias_idx), B, H, W, ws, heads, dh, noff, ctypes.c_float(dh ** -0.5), out.ptr, probs.ptr, wsf.ptr)
    out.check("attn_fwd out")
    probs.check("attn_fwd probs")
    wsf.check("attn_fwd workspace", fully_written=False)
    dqkv, dS = Guarded(qkv.numel()), Guarded(pe)
    _lib.call("suhpe_window_attn_bwd", P(qkv), probs.ptr, P(dout), P(bias_idx), B, H, W, ws, heads, dh, noff, ctypes.c_float(dh ** -0.5), dqkv.ptr, None, dS.ptr)
    dqkv.check("attn_bwd dqkv")
    dS.check("attn_bwd score gradient")
    # inverse of bias_idx in CSR form, then the reproducible bias-table gradient
    flat = idx.reshape(-1)
    order = torch.argsort(flat, stable=True)
    inv_list = order.to(torch.int32).cuda().contiguous()
    inv_ptr = torch.zeros(noff + 1, dtype=torch.int32)
    inv_ptr[1:] = torch.cumsum(torch.bincount(flat, minlength=noff), 0).to(torch.int32)
    inv_ptr = inv_ptr.cuda()
    dtab, wsd = Guarded(heads * noff), Guarded(max(1, lib.suhpe_window_attn_dbias_workspace_bytes(ws, heads) // 4))
    dtab.payload.zero_()
    _lib.call("suhpe_window_attn_dbias", dS.ptr, P(inv_ptr), P(inv_list), B, H, W, ws, heads, noff, dtab.ptr, wsd.ptr)
    dtab.check("attn_dbias table")
    wsd.check("attn_dbias workspace", fully_written=False)
    # the probabilities of a window are a softmax: rows sum to one
    pr = probs.payload.view(-1, N, N)
    assert float((pr.sum(-1) - 1.0).abs().max()) < 1e-4


@pytest.mark.parametrize("cfg", ATTN_CASES, ids=str)
def test_recomputing_window_attention_stays_inside_its_buffers(cfg):
    """suhpe_window_attn_fwd_rs / _bwd_rs: output, row statistics, the two tile-ordered bias expansions, dqkv, the score-gradient workspace"""
    _lib, lib, _ = _lib_and_names()
    from semiuhpe_amd.src.tiny_vit import attention_bias_index
    for f in ("suhpe_window_attn_rs_workspace_bytes", "suhpe_window_attn_rowstats_elems", "suhpe_window_attn_probs_elems"):
        getattr(lib, f).restype = ctypes.c_int64
    B, H, W, ws, heads = cfg
    dh = 32
    gen = torch.Generator().manual_seed(sum(cfg) + 1)
    idx, noff = attention_bias_index(ws)
    bias_idx = idx.to(torch.int32).cuda().contiguous()
    qkv = torch.randn(B * H * W, heads * 3 * dh, generator=gen).cuda()
    table = torch.randn(heads, noff, generator=gen).cuda()
    dout = torch.randn(B * H * W, heads * dh, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    np_ = 64 if ws * ws <= 64 else 224
    assert lib.suhpe_window_attn_rowstats_elems(B, H, W, ws, heads) == B * (H // ws) * (W // ws) * heads * np_ * 2
    assert lib.suhpe_window_attn_rs_workspace_bytes(ws, heads) == 2 * heads * np_ * np_ * 4
    out, stats = Guarded(B * H * W * heads * dh), Guarded(lib.suhpe_window_attn_rowstats_elems(B, H, W, ws, heads))
    wsf = Guarded(lib.suhpe_window_attn_rs_workspace_bytes(ws, heads) // 4)
    _lib.call("suhpe_window_attn_fwd_rs", P(qkv), P(table), P(bias_idx), B, H, W, ws, heads, dh, noff, ctypes.c_float(dh ** -0.5), out.ptr, stats.ptr, wsf.ptr)
    out.check("attn_fwd_rs out"); stats.check("attn_fwd_rs row statistics"); wsf.check("attn_fwd_rs bias tiles")
    dqkv, dS, wsb = Guarded(qkv.numel()), Guarded(lib.suhpe_window_attn_probs_elems(B, H, W, ws, heads)), Guarded(lib.suhpe_window_attn_rs_workspace_bytes(ws, heads) // 4)
    dtab = Guarded(heads * noff)
    dtab.payload.zero_()
    _lib.call("suhpe_window_attn_bwd_rs", P(qkv), out.ptr, stats.ptr, P(dout), P(table), P(bias_idx), B, H, W, ws, heads, dh, noff, ctypes.c_float(dh ** -0.5),
              dqkv.ptr, dtab.ptr, dS.ptr, wsb.ptr)
    dqkv.check("attn_bwd_rs dqkv"); dS.check("attn_bwd_rs score gradient"); wsb.check("attn_bwd_rs bias tiles"); dtab.check("attn_bwd_rs bias-table gradient")
    # inference form: no statistics pointer
    out2 = Guarded(B * H * W * heads * dh)
    _lib.call("suhpe_window_attn_fwd_rs", P(qkv), P(table), P(bias_idx), B, H, W, ws, heads, dh, noff, ctypes.c_float(dh ** -0.5), out2.ptr, None, wsf.ptr)
    out2.check("attn_fwd_rs out (inference)")
    assert torch.equal(out.payload, out2.payload)


@pytest.mark.parametrize("rows,c", [(784 * 3, 192), (49 * 5 + 3, 576), (1000, 100), (37, 384)], ids=str)
def test_layernorm_stays_inside_its_buffers(rows, c):
    _lib, lib, _ = _lib_and_names()
    lib.suhpe_bn_workspace_bytes.restype = ctypes.c_int64
    gen = torch.Generator().manual_seed(rows + c)
    x, dy = torch.randn(rows, c, generator=gen).cuda(), torch.randn(rows, c, generator=gen).cuda()
    gamma, beta = torch.randn(c, generator=gen).cuda(), torch.randn(c, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    y, mean, rstd = Guarded(rows * c), Guarded(rows), Guarded(rows)
    _lib.call("suhpe_layernorm_fwd", P(x), ctypes.c_int64(rows), c, ctypes.c_float(1e-5), P(gamma), P(beta), y.ptr, mean.ptr, rstd.ptr)
    for gb, what in ((y, "y"), (mean, "mean"), (rstd, "rstd")):
        gb.check("layernorm_fwd " + what)
    ref = torch.nn.functional.layer_norm(x.double().cpu(), (c,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
    assert float((y.payload.cpu().double() - ref.reshape(-1)).abs().max()) < 1e-4
    ws, dx, dg, db = Guarded(lib.suhpe_bn_workspace_bytes(c) // 4), Guarded(rows * c), Guarded(c), Guarded(c)
    _lib.call("suhpe_layernorm_bwd_skip", P(x), P(dy), ctypes.c_int64(rows), c, mean.ptr, rstd.ptr, P(gamma), P(x), dx.ptr, dg.ptr, db.ptr, 0, ws.ptr)
    ws.check("layernorm_bwd workspace", fully_written=False)
    for gb, what in ((dx, "dx"), (dg, "dgamma"), (db, "dbeta")):
        gb.check("layernorm_bwd " + what)


@pytest.mark.parametrize("n,c,csq,hw", [(7, 96, 8, 49), (33, 512, 32, 196), (5, 1536, 64, 49), (160, 960, 40, 4)], ids=str)
def test_squeeze_excitation_section_stays_inside_its_buffers(n, c, csq, hw):
    """depthwise BatchNorm -> SiLU -> squeeze, excitation MLP (forward / backward with its workspace), scale pass, backward reduce / combine / apply"""
    _lib, lib, _ = _lib_and_names()
    gen = torch.Generator().manual_seed(n + c)
    rows = n * hw
    x, dout = torch.randn(rows, c, generator=gen).cuda(), torch.randn(rows, c, generator=gen).cuda()
    mean, rstd = torch.randn(c, generator=gen).cuda() * 0.1, (torch.rand(c, generator=gen) + 0.5).cuda()
    gamma, beta = torch.randn(c, generator=gen).cuda(), torch.randn(c, generator=gen).cuda()
    w1, b1 = (torch.randn(csq, c, generator=gen) * 0.05).cuda(), torch.randn(csq, generator=gen).cuda()
    w2, b2 = (torch.randn(c, csq, generator=gen) * 0.05).cuda(), torch.randn(c, generator=gen).cuda()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    pool = Guarded(n * c)
    _lib.call("suhpe_bn_act_pool_fwd", P(x), ctypes.c_int64(rows), ctypes.c_int64(0), c, hw, P(mean), P(rstd), P(gamma), P(beta), 3, None, pool.ptr)
    pool.check("bn_act_pool_fwd (squeeze alone)")
    a1, s = Guarded(n * csq), Guarded(n * c)
    _lib.call("suhpe_se_mlp_fwd", pool.ptr, P(w1), P(b1), P(w2), P(b2), n, c, csq, a1.ptr, s.ptr)
    a1.check("se_mlp_fwd a1"); s.check("se_mlp_fwd s")
    y = Guarded(rows * c)
    _lib.call("suhpe_bn_act_scale_fwd", P(x), ctypes.c_int64(rows), ctypes.c_int64(0), c, hw, P(mean), P(rstd), P(gamma), P(beta), 3, s.ptr, y.ptr)
    y.check("bn_act_scale_fwd")
    sums = Guarded(5 * n * c)
    _lib.call("suhpe_se_bn_bwd_reduce", P(x), P(dout), ctypes.c_int64(rows), ctypes.c_int64(0), c, hw, P(mean), P(rstd), P(gamma), P(beta), 3, sums.ptr)
    sums.check("se_bn_bwd_reduce sums")
    ws, dp = Guarded(n * (c + 2 * csq)), Guarded(n * c)
    dws = [Guarded(t.numel()) for t in (w1, b1, w2, b2)]
    for g_ in dws:
        g_.payload.zero_()
    _lib.call("suhpe_se_mlp_bwd", sums.ptr, s.ptr, a1.ptr, pool.ptr, P(w1), P(w2), n, c, csq, dp.ptr, dws[0].ptr, dws[1].ptr, dws[2].ptr, dws[3].ptr, ws.ptr)
    ws.check("se_mlp_bwd workspace", fully_written=False)
    dp.check("se_mlp_bwd dp")
    for g_, what in zip(dws, ("dw1", "db1", "dw2", "db2")):
        g_.check("se_mlp_bwd " + what)
    chan, dg, db = Guarded(4 * c), Guarded(c), Guarded(c)
    _lib.call("suhpe_se_bn_bwd_combine", sums.ptr, s.ptr, dp.ptr, n, 0, c, hw, chan.ptr, dg.ptr, db.ptr, 0)
    chan.check("se_bn_bwd_combine chan", fully_written=False)
    dg.check("se_bn_bwd_combine dgamma"); db.check("se_bn_bwd_combine dbeta")
    dyo = Guarded(rows * c)
    _lib.call("suhpe_se_bn_bwd_apply", P(x), P(dout), ctypes.c_int64(rows), ctypes.c_int64(0), c, hw, P(mean), P(rstd), P(gamma), P(beta), 3, s.ptr, dp.ptr, chan.ptr,
              1, dyo.ptr)
    dyo.check("se_bn_bwd_apply")


@pytest.mark.parametrize("n,hw,c,seg,nb", [(6, 196, 512, 2, 3), (5, 49, 2048, 0, 2), (4, 25, 6, 3, 3), (3, 784, 132, 1, 3)], ids=str)
def test_repvgg_block_tail_stays_inside_its_buffers(n, hw, c, seg, nb):
    """suhpe_repvgg_bn_fwd / _fwd_stats: the output and the statistics records (exactly suhpe_repvgg_bn_fwd_stats_blocks() of them, every one written)"""
    _lib, lib, _ = _lib_and_names()
    gen = torch.Generator().manual_seed(n * hw + c)
    rows, rows_a = n * hw, seg * hw
    xs = [torch.randn(rows, c, generator=gen).cuda() for _ in range(nb)]
    k = 2 if seg else 1
    mean = [torch.randn(k, c, generator=gen).cuda() * 0.1 for _ in range(nb)]
    rstd = [(torch.rand(k, c, generator=gen) + 0.5).cuda() for _ in range(nb)]
    gam = [(torch.rand(c, generator=gen) + 0.5).cuda() for _ in range(nb)]
    bet = [torch.randn(c, generator=gen).cuda() for _ in range(nb)]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    first = ctypes.c_int32(0)
    tiles = lib.suhpe_repvgg_bn_fwd_stats_blocks(ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, ctypes.byref(first))
    assert tiles > 0 and (0 < first.value < tiles if seg else first.value == tiles)
    y0, y1, st = Guarded(rows * c), Guarded(rows * c), Guarded(tiles * c * 4)
    _lib.call("suhpe_repvgg_bn_fwd", arr(xs), arr(mean), arr(rstd), arr(gam), arr(bet), nb, ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, y0.ptr)
    y0.check("repvgg_bn_fwd y")
    _lib.call("suhpe_repvgg_bn_fwd_stats", arr(xs), arr(mean), arr(rstd), arr(gam), arr(bet), nb, ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, y1.ptr, st.ptr)
    y1.check("repvgg_bn_fwd_stats y"); st.check("repvgg_bn_fwd_stats records")
    assert torch.equal(y0.payload, y1.payload)
    rec = st.payload.view(tiles, c, 4)
    assert float(rec[:first.value, :, 3].sum(0).min()) == float(rec[:first.value, :, 3].sum(0).max()) == float(rows_a if seg else rows)     # the counts add up to the rows
    # the backward, mask from y and mask recomputed: three input gradients each, the workspace of suhpe_bn_workspace_bytes(c)
    lib.suhpe_bn_workspace_bytes.restype = ctypes.c_int64
    dout = torch.randn(rows, c, generator=gen).cuda()
    marr = lambda gs: (ctypes.c_void_p * len(gs))(*[g_.ptr.value for g_ in gs])
    for remask in (False, True):
        dxs = [Guarded(rows * c) for _ in range(nb)]
        dgs, dbs = [Guarded(c) for _ in range(nb)], [Guarded(c) for _ in range(nb)]
        for g_ in dgs + dbs:
            g_.payload.zero_()
        ws = Guarded(lib.suhpe_bn_workspace_bytes(c) // 4)
        if remask:
            _lib.call("suhpe_repvgg_bn_bwd_remask", arr(xs), arr(mean), arr(rstd), arr(gam), arr(bet), nb, P(dout), ctypes.c_int64(rows), ctypes.c_int64(rows_a), c,
                      marr(dxs), marr(dgs), marr(dbs), ws.ptr)
        else:
            _lib.call("suhpe_repvgg_bn_bwd", arr(xs), arr(mean), arr(rstd), arr(gam), nb, P(dout), y0.ptr, ctypes.c_int64(rows), ctypes.c_int64(rows_a), c,
                      marr(dxs), marr(dgs), marr(dbs), ws.ptr)
        for j in range(nb):
            dxs[j].check(f"repvgg_bn_bwd dx[{j}] remask={remask}"); dgs[j].check("repvgg_bn_bwd dgamma"); dbs[j].check("repvgg_bn_bwd dbeta")
        ws.check("repvgg_bn_bwd workspace", fully_written=False)
