"""Winograd F(2x2, 3x3) tile against the measured best direct tile on the 3x3 / stride 1 shapes of C2 (EfficientNetV2-S) and C3 (RepVGG-B1g2):
    python tools/wino_bench.py [c2|c3|all] [--check-only]
per (shape, pass): max error of the Winograd result against a float64 torch reference next to the direct tile's own error, the
BatchNorm statistics from the epilogue records (both segment layouts), and cold-cache times of both tiles.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from semiuhpe_amd import _lib, kernels as K

lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
WINO = names.index("wino_nt_kernel<64,64>")
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")

# (cin, cout, groups, hw, batch fwd, seg, label)
C2 = [(24, 24, 1, 112, 160, 32, "c2 s1"), (48, 192, 1, 56, 160, 32, "c2 s2"), (64, 256, 1, 28, 160, 32, "c2 s3")]
C3 = [(128, 128, 1, 56, 320, 64, "c3 st1"), (128, 128, 2, 56, 320, 64, "c3 st1 g2"), (256, 256, 1, 28, 320, 64, "c3 st2"), (256, 256, 2, 28, 320, 64, "c3 st2 g2"),
      (512, 512, 1, 14, 320, 64, "c3 st3"), (512, 512, 2, 14, 320, 64, "c3 st3 g2")]
SMALL = [(8, 12, 1, 6, 3, 1, "tiny"), (20, 36, 1, 10, 5, 2, "odd tiles"), (16, 16, 2, 4, 7, 3, "groups")]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3


def pin(d, pass_id, tile):
    rc = lib.suhpe_conv2d_set_tile(ctypes.byref(d), pass_id, tile)
    assert rc == 0, (rc, tile)


def run(shapes, check_only=False, big_ref=True):
    for cin, cout, g, hw, n, seg, label in shapes:
        torch.manual_seed(0)
        x = torch.randn(n, hw, hw, cin, device="cuda")
        w_t = torch.randn(cout, cin // g, 3, 3, device="cuda") * (2.0 / (9 * cin // g)) ** 0.5
        w = K.to_filter_layout(w_t)
        d = K.conv_desc(x.shape, w.shape, 1, 1, g)
        fl = K.conv_flops(d)
        dy = torch.randn(n, hw, hw, cout, device="cuda")
        skip = torch.randn_like(x)
        # float64 reference on a slice of the batch (first / last images)
        ns = min(n, 4)
        idx = list(range(ns // 2)) + list(range(n - (ns - ns // 2), n))
        xr = x[idx].cpu().double().permute(0, 3, 1, 2)
        yr = F.conv2d(xr, w_t.cpu().double(), None, 1, 1, 1, g).permute(0, 2, 3, 1).cuda()
        dyr = dy[idx].cpu().double().permute(0, 3, 1, 2)
        dxr = F.conv_transpose2d(dyr, w_t.cpu().double(), None, 1, 1, 0, g).permute(0, 2, 3, 1).cuda() + skip[idx].double()
        res = {}
        # the direct side: the fastest of the direct 3x3 candidates per pass (the tile database pins the Winograd tile for most of these shapes by now)
        best_direct = [-1, -1]
        if not check_only:
            cands = [i for i, nm in enumerate(names) if nm.startswith("igemm_nt3x3") or nm in ("igemm_nt_kernel<128,64,32,32,4>", "igemm_nt_kernel<128,32,32,32,4>")]
            for pid, fn in ((0, lambda: K.conv2d_fwd(x, w, None, 1, 1, g)), (1, lambda: K.conv2d_dgrad(dy, w, x.shape, 1, 1, g, skip=skip))):
                bt = 1e9
                for c in cands:
                    if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, c) != 0:
                        continue
                    tm = timeit(fn, 3)
                    if tm < bt:
                        bt, best_direct[pid] = tm, c
        for tag, tile in (("direct", None), ("wino", WINO)):
            if tile is None:
                pin(d, 0, best_direct[0] if best_direct[0] >= 0 else names.index("igemm_nt_kernel<128,64,32,32,4>"))
                pin(d, 1, best_direct[1] if best_direct[1] >= 0 else names.index("igemm_nt_kernel<128,64,32,32,4>"))
            else:
                pin(d, 0, tile); pin(d, 1, tile)
            y = K.conv2d_fwd(x, w, None, 1, 1, g)
            dx = K.conv2d_dgrad(dy, w, x.shape, 1, 1, g, skip=skip)
            ys, part, tiles, (first, extra) = K.conv2d_fwd_stats(x, w, 1, 1, g, seg=seg)
            mean, rstd = K.bn_stats_from_partials_seg(part, tiles, first, extra, n * hw * hw, seg * hw * hw, cout, 1e-3, 0.1)
            ys1, part1, tiles1 = K.conv2d_fwd_stats(x, w, 1, 1, g)
            mean1, rstd1 = K.bn_stats_from_partials(part1, tiles1, n * hw * hw, cout, 1e-3, 0.1)
            res[tag] = dict(y=y, dx=dx, ys=ys, mean=mean, rstd=rstd, mean1=mean1, rstd1=rstd1)
            e_y = (y[idx].double() - yr).abs().max().item() / yr.abs().max().item()
            e_dx = (dx[idx].double() - dxr).abs().max().item() / dxr.abs().max().item()
            res[tag]["err"] = (e_y, e_dx)
            if not check_only:
                res[tag]["t"] = (timeit(lambda: K.conv2d_fwd(x, w, None, 1, 1, g)), timeit(lambda: K.conv2d_dgrad(dy, w, x.shape, 1, 1, g, skip=skip)),
                                 timeit(lambda: K.conv2d_fwd_stats(x, w, 1, 1, g, seg=seg)))
        # filter gradient: Winograd F(3x3, 2x2) tile against the direct tile (float64 reference on the CPU for the small shapes)
        tn_names = [lib.suhpe_conv2d_tile_name(2, i).decode() for i in range(lib.suhpe_conv2d_tile_count(2))]
        pin(d, 2, -1)
        dw_a = K.conv2d_wgrad(x, dy, w.shape, 1, 1, g)
        pin(d, 2, tn_names.index("wino_wgrad_kernel<64,64>"))
        dw_b = K.conv2d_wgrad(x, dy, w.shape, 1, 1, g)
        wg_rel = (dw_a - dw_b).abs().max().item() / dw_a.abs().max().item()
        wg64 = ""
        if n * hw * hw * cin * cout <= 4e8:
            xx = x.cpu().double().permute(0, 3, 1, 2).requires_grad_(False)
            wref = torch.zeros(cout, cin // g, 3, 3, dtype=torch.float64, requires_grad=True)
            F.conv2d(xx, wref, None, 1, 1, 1, g).backward(dy.cpu().double().permute(0, 3, 1, 2))
            ref = wref.grad.cuda()
            la, lb = K.from_filter_layout(dw_a) if hasattr(K, "from_filter_layout") else dw_a, K.from_filter_layout(dw_b) if hasattr(K, "from_filter_layout") else dw_b
            wg64 = f" (vs fp64: direct {(la.double() - ref).abs().max().item() / ref.abs().max().item():.1e} wino {(lb.double() - ref).abs().max().item() / ref.abs().max().item():.1e})"
        t_wg = ""
        if not check_only:
            tb_ = timeit(lambda: K.conv2d_wgrad(x, dy, w.shape, 1, 1, g))
            best = 1e9
            for c in range(len(tn_names) - 1):
                if lib.suhpe_conv2d_set_tile(ctypes.byref(d), 2, c) == 0:
                    best = min(best, timeit(lambda: K.conv2d_wgrad(x, dy, w.shape, 1, 1, g), 3))
            t_wg = f" wgrad {best*1e6:.0f} -> {tb_*1e6:.0f} us ({fl/best/1e12:.0f} -> {fl/tb_/1e12:.0f} TF eff)"
        pin(d, 2, -1)
        # the self-transforming form of the kernel (no Winograd-domain filter handed over): plain entry points with the tile pinned
        from semiuhpe_amd._lib import call, ptr
        y_self = torch.empty_like(res["wino"]["y"])
        call("suhpe_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w), None, ptr(y_self))
        self_err = (y_self - res["wino"]["y"]).abs().max().item() / res["wino"]["y"].abs().max().item()
        pin(d, 0, -1); pin(d, 1, -1)
        a, b = res["direct"], res["wino"]
        ymax = a["y"].abs().max().item()
        # statistics against the tensors themselves (float64)
        yy = b["ys"].double().reshape(n, -1, cout)
        m_a, m_b = yy[:seg].mean((0, 1)), yy[seg:].mean((0, 1))
        v_a, v_b = yy[:seg].var((0, 1), unbiased=False), yy[seg:].var((0, 1), unbiased=False)
        st_err = max((b["mean"][0].double() - m_a).abs().max().item(), (b["mean"][1].double() - m_b).abs().max().item(),
                     ((b["rstd"][0].double() - (v_a + 1e-3).rsqrt()) / (v_a + 1e-3).rsqrt()).abs().max().item(),
                     ((b["rstd"][1].double() - (v_b + 1e-3).rsqrt()) / (v_b + 1e-3).rsqrt()).abs().max().item())
        yall = yy.reshape(-1, cout)
        st1_err = max((b["mean1"].double() - yall.mean(0)).abs().max().item(),
                      ((b["rstd1"].double() - (yall.var(0, unbiased=False) + 1e-3).rsqrt()) * (yall.var(0, unbiased=False) + 1e-3).sqrt()).abs().max().item())
        line = (f"{label:11s} {cin}->{cout} g{g} {hw}x{hw} n{n}: rel err vs fp64 fwd direct {a['err'][0]:.1e} wino {b['err'][0]:.1e} | dgrad+skip direct {a['err'][1]:.1e} "
                f"wino {b['err'][1]:.1e} | wino-direct max {((a['y'] - b['y']).abs().max().item() / ymax):.1e} stats_y==y {torch.equal(b['ys'], b['y'])} "
                f"stats err seg {st_err:.1e} one {st1_err:.1e} self-vs-pre {self_err:.1e} | wgrad wino-direct {wg_rel:.1e}{wg64}{t_wg}")
        if not check_only:
            ta, tb = a["t"], b["t"]
            line += (f" | fwd {ta[0]*1e6:.0f} -> {tb[0]*1e6:.0f} us ({fl/ta[0]/1e12:.0f} -> {fl/tb[0]/1e12:.0f} TF eff), dgrad {ta[1]*1e6:.0f} -> {tb[1]*1e6:.0f} us "
                     f"({fl/ta[1]/1e12:.0f} -> {fl/tb[1]/1e12:.0f}), fwd+stats {ta[2]*1e6:.0f} -> {tb[2]*1e6:.0f} us")
        print(line, flush=True)


if __name__ == "__main__" and os.environ.get("WINO_SHAPES"):       # WINO_SHAPES="cin,cout,groups,hw,n,seg;..."
    run([tuple(int(v) for v in sh.split(",")) + (sh,) for sh in os.environ["WINO_SHAPES"].split(";")])
    sys.exit(0)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    check_only = "--check-only" in sys.argv
    run(SMALL, check_only=True)
    if which in ("c2", "all"):
        run(C2, check_only)
    if which in ("c3", "all"):
        run(C3, check_only)
