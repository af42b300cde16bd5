import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
lib = _lib.load(); lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, t) for t in range(lib.suhpe_conv2d_tile_count(0))]
narrow, igemm = names.index(b"gemm_nt_dma<256,64>"), names.index(b"igemm_nt_kernel<64,64,32,32,4>")
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
def cold_min(fn, reps=7):
    fn(); best = 1e9
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3
for (n, h, cin, cout) in ((160, 56, 192, 48), (160, 56, 192, 64), (160, 28, 256, 64), (160, 56, 96, 48)):
    x = torch.randn(n, h, h, cin, device="cuda")
    wt = K.to_filter_layout(torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5)
    d = K.conv_desc(x.shape, wt.shape, 1, 0, 1)
    sc, sh = torch.rand(2, cin, device="cuda") + 0.5, torch.randn(2, cin, device="cuda") * 0.3
    sc1, sh1 = sc[:1].contiguous(), sh[:1].contiguous()
    for name, tile in (("igemm", igemm), ("narrow", narrow)):
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, tile)
        t_plain = cold_min(lambda: K.conv2d_fwd(x, wt, None, 1, 0, 1))
        t_st = cold_min(lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, seg=32))
        t_xf = cold_min(lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, seg=32, xf=(sc, sh, None, "silu", h * h, 32 * h * h)))
        t_xf1 = cold_min(lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, xf=(sc1, sh1, None, "silu", h * h, 0)))
        t_xfn = cold_min(lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, xf=(sc1, sh1, None, "none", h * h, 0)))
        print(f"{n}x{h}x{h} {cin}->{cout} {name:7s} plain {t_plain:6.1f}  +stats {t_st:6.1f}  +xf silu 2seg {t_xf:6.1f}  xf silu 1seg {t_xf1:6.1f}  xf none {t_xfn:6.1f} us", flush=True)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
