import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
lib = _lib.load(); lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, t) for t in range(lib.suhpe_conv2d_tile_count(0))]
narrow, igemm = names.index(b"gemm_nt_dma<256,64>"), names.index(b"igemm_nt_kernel<64,64,32,32,4>")
n, cin, cout, h, w, seg, act = 20, 192, 48, 28, 28, 4, "silu"
g = torch.Generator().manual_seed(53)
x = (torch.randn(n, h, w, cin, generator=g) * 1.3 + 0.2).cuda()
wg = K.to_filter_layout((torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).cuda())
scale, shift = (torch.rand(2, cin, generator=g) + 0.5).cuda(), (torch.randn(2, cin, generator=g) * 0.3).cuda()
d = K.conv_desc(x.shape, wg.shape, 1, 0, 1)
rows, hw = n * h * w, h * w
rows_a = seg * hw
res = {}
for name, tile in (("igemm", igemm), ("narrow", narrow)):
    assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, tile) == 0
    for form in ("xf", "plain"):
        xf = (scale, shift, None, act, hw, rows_a) if form == "xf" else None
        y, part, nt, (first, extra) = K.conv2d_fwd_stats(x, wg, 1, 0, 1, seg=seg, xf=xf)
        st = K.bn_stats_from_partials_seg(part, nt, first, extra, rows, rows_a, cout, 1e-3, 0.1, consume=False)
        yd = y.double().cpu().reshape(rows, cout)
        want_m = torch.stack([yd[:rows_a].mean(0), yd[rows_a:].mean(0)])
        want_r = torch.stack([(yd[:rows_a].var(0, unbiased=False) + 1e-3).rsqrt(), (yd[rows_a:].var(0, unbiased=False) + 1e-3).rsqrt()])
        print(name, form, "tiles", nt, "first", first, "extra", extra is not None, "mean err", float((st[0].double().cpu() - want_m).abs().max()),
              "rstd err", float((st[1].double().cpu() - want_r).abs().max()), "rec n sum", float(part.reshape(-1, cout, 4)[:nt, 0, 3].sum()))
        res[(name, form)] = (y, st)
for form in ("xf", "plain"):
    a, b = res[("igemm", form)], res[("narrow", form)]
    print(form, "y maxdiff", float((a[0] - b[0]).abs().max()), "mean diff", float((a[1][0] - b[1][0]).abs().max()), "rstd diff", float((a[1][1] - b[1][1]).abs().max()))
assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, narrow) == 0
xf = (scale, shift, None, act, hw, rows_a)
y, part, nt, (first, extra) = K.conv2d_fwd_stats(x, wg, 1, 0, 1, seg=seg, xf=xf)
torch.cuda.synchronize()
r = part.reshape(-1, cout, 4)[:nt].cpu()
ok = (r[:, :, 3] == 32).all(1)
print("records with n == 32 for every channel:", int(ok.sum()), "of", nt)
idx = torch.nonzero(ok).flatten().tolist()
print("good units:", idx[:40], "...", idx[-10:])
bad = torch.nonzero(~ok).flatten().tolist()
print("bad units:", bad[:40])
print("a bad record:", r[bad[0], :4] if bad else None)
