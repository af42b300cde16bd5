"""One-off stress of the Winograd kernels' hand-counted waits: many repetitions on bench-sized shapes while other streams saturate HBM and L2; every result
bit-identical to the quiet run.   python tools/debug/wino_stress.py [iterations]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = {p: [lib.suhpe_conv2d_tile_name(p, i).decode() for i in range(lib.suhpe_conv2d_tile_count(p))] for p in (0, 1, 2)}
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gen = torch.Generator().manual_seed(3)
bad = 0
for n, cin, cout, hw, g in ((160, 48, 192, 56, 1), (160, 64, 256, 28, 1), (128, 24, 24, 112, 1), (320, 512, 512, 14, 2), (160, 192, 48, 56, 1), (64, 256, 256, 28, 1)):
    x = torch.randn(n, hw, hw, cin, generator=gen).cuda()
    dy = torch.randn(n, hw, hw, cout, generator=gen).cuda()
    skip = torch.randn(n, hw, hw, cin, generator=gen).cuda()
    wg = K.to_filter_layout((torch.randn(cout, cin // g, 3, 3, generator=gen) / (3 * (cin // g) ** 0.5)).cuda())
    sc, sh = torch.rand(cout, generator=gen).cuda() + 0.5, torch.randn(cout, generator=gen).cuda()
    d = K.conv_desc(x.shape, wg.shape, 1, 1, g)
    for pid, nm in ((0, "wino_nt_kernel<64,64>"), (1, "wino_nt_kernel<64,64>"), (2, "wino_wgrad_kernel<64,64>")):
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, names[pid].index(nm)) == 0
    ops = {"fwd": lambda: K.conv2d_fwd(x, wg, None, 1, 1, g), "dgrad+skip": lambda: K.conv2d_dgrad(dy, wg, x.shape, 1, 1, g, skip=skip),
           "wgrad": lambda: K.conv2d_wgrad(x, dy, wg.shape, 1, 1, g), "stats": lambda: (lambda r: r[1][: cout * r[2] * 4].clone())(K.conv2d_fwd_stats(x, wg, 1, 1, g, seg=n // 5)),      # (the written records: the boundary record exists only when the segment cuts a 16-tile block)
           "affine": lambda: K.conv2d_fwd_affine(x, wg, sc, sh, "silu", None, False, 1, 1, g)}
    quiet = {k: f() for k, f in ops.items()}
    torch.cuda.synchronize()
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    small = torch.empty(1 << 22, dtype=torch.float32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        for _ in range(iters * 3):
            big[: 1 << 27].copy_(big[1 << 27:], non_blocking=True)
    with torch.cuda.stream(s2):
        for _ in range(iters * 40):
            small.mul_(1.0001)                     # an L2-resident elementwise kernel: CU and L2 pressure between the launches
    for it in range(iters):
        for k, f in ops.items():
            if not torch.equal(f(), quiet[k]):
                bad += 1
                print("MISMATCH", (n, cin, cout, hw, g), k, it, flush=True)
    torch.cuda.synchronize()
    for pid in (0, 1, 2):
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    print((n, cin, cout, hw, g), "done", flush=True)
    del big, small
print("mismatches:", bad)
