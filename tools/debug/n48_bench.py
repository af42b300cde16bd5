"""input gradient of the 48 -> 192 3x3 FusedMBConv expand conv (N = 48 output columns): the 48-column halo tile against the 64-column ones"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400000)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3
for n, hw, cin, cout in ((160, 56, 48, 192), (160, 56, 24, 96)):
    x_shape = (n, hw, hw, cin)
    dy = torch.randn(n, hw, hw, cout, device="cuda"); skip = torch.randn(*x_shape, device="cuda")
    w = K.to_filter_layout(torch.randn(cout, cin, 3, 3, device="cuda") / 20)
    d = K.conv_desc(x_shape, w.shape, 1, 1, 1)
    K.conv2d_dgrad(dy, w, x_shape, 1, 1, 1, skip=skip)
    flops = 2.0 * n * hw * hw * cin * cout * 9
    for pin in ("igemm_nt3x3_kernel<64,64,32,32>", "igemm_nt3x3_kernel<128,64,32,32>", "igemm_nt3x3_kernel<128,32,32,32>", "igemm_nt_kernel<128,64,32,32,4>", "igemm_nt3x3_n48_kernel<128,48>"):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, names.index(pin)) != 0:
            continue
        us = timed(lambda: K.conv2d_dgrad(dy, w, x_shape, 1, 1, 1, skip=skip))
        print(f"{n}x{hw}x{hw} {cin}<-{cout} dgrad+skip  {pin:40s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s", flush=True)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), 1, -1)
print("forward (no statistics epilogue):")
for n in (160, 128):
    hw, cin, cout = 56, 48, 192
    x = torch.randn(n, hw, hw, cin, device="cuda")
    w = K.to_filter_layout(torch.randn(cout, cin, 3, 3, device="cuda") / 20)
    d = K.conv_desc(x.shape, w.shape, 1, 1, 1)
    flops = 2.0 * n * hw * hw * cin * cout * 9
    for pin in ("igemm_nt3x3_kernel<64,64,32,32>", "igemm_nt3x3_kernel<128,64,32,32>", "igemm_nt_kernel<128,64,32,32,4>", "igemm_nt_kernel<64,64,32,32,4>", "igemm_nt3x3_n48_kernel<128,48>"):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names.index(pin)) != 0:
            continue
        us = timed(lambda: K.conv2d_fwd(x, w, None, 1, 1, 1))
        print(f"{n}x{hw}x{hw} {cin}->{cout} fwd  {pin:40s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s", flush=True)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
