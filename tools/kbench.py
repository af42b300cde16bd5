"""Per-shape micro-benchmark of the conv / GEMM kernels on the layer shapes of a backbone (run on the GPU box).

    python tools/kbench.py [effnet|repvgg] [batch]

Prints, per distinct (shape, pass): kernel instance, time, TFLOP/s, and algorithmic GB/s.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from semiuhpe_amd import _lib, kernels as K  # noqa: E402


def effnet_convs():
    """(cin, cout, k, stride, groups, in_hw, count) for EfficientNetV2-S at 224x224"""
    out = [(3, 24, 3, 2, 1, 224, 1)]
    from semiuhpe_amd.src.efficientnet_v2 import V2_S
    hw = 112
    for kind, e, k, s, cin, cout, n in V2_S:
        for i in range(n):
            ci, st = (cin, s) if i == 0 else (cout, 1)
            exp = ci * e
            if kind == "fused":
                if exp != ci:
                    out.append((ci, exp, 3, st, 1, hw, 1)); out.append((exp, cout, 1, 1, 1, hw // st, 1))
                else:
                    out.append((ci, cout, 3, st, 1, hw, 1))
            else:
                out.append((ci, exp, 1, 1, 1, hw, 1)); out.append((exp, exp, 3, st, exp, hw, 1)); out.append((exp, cout, 1, 1, 1, hw // st, 1))
            hw //= st
    out.append((256, 1280, 1, 1, 1, 7, 1))
    agg = {}
    for c in out:
        agg[c[:6]] = agg.get(c[:6], 0) + 1
    return [k + (v,) for k, v in agg.items()]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    lib = _lib.load()
    tot = {0: 0.0, 1: 0.0, 2: 0.0}
    totf = 0.0
    print(f"{'shape':42s} {'pass':5s} {'kernel':38s} {'us':>9s} {'TF/s':>7s} {'GB/s':>8s}  x count")
    for cin, cout, k, st, g, hw, cnt in effnet_convs():
        x = torch.randn(batch, hw, hw, cin, device="cuda")
        w = K.to_filter_layout(torch.randn(cout, cin // g, k, k, device="cuda"))
        pad = (k - 1) // 2
        d = K.conv_desc(x.shape, w.shape, st, pad, g)
        y = K.conv2d_fwd(x, w, None, st, pad, g)
        dy = torch.randn_like(y)
        flops, nbytes = K.conv_flops(d), K.conv_bytes(d)
        for pid, name, fn in ((0, "fwd", lambda: K.conv2d_fwd(x, w, None, st, pad, g)),
                              (1, "dgrad", lambda: K.conv2d_dgrad(dy, w, x.shape, st, pad, g)),
                              (2, "wgrad", lambda: K.conv2d_wgrad(x, dy, w.shape, st, pad, g))):
            if pid == 1 and cin == 3:
                continue
            t = timeit(fn)
            kn = "dwconv" if g == cin and g > 1 else lib.suhpe_conv2d_kernel_name(ctypes.byref(d), pid).decode()
            print(f"{str((cin, cout, k, st, g, hw)):42s} {name:5s} {kn:38s} {t*1e6:9.1f} {flops/t/1e12:7.2f} {nbytes/t/1e9:8.1f}  x{cnt}")
            tot[pid] += t * cnt
        totf += flops * cnt
    print("per-image-batch totals (s): fwd %.4f dgrad %.4f wgrad %.4f ; fwd TF/s overall %.2f" % (tot[0], tot[1], tot[2], totf / tot[0] / 1e12))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("gemm", "ksweep")):
    main()


def gemm_main():
    """asymptotic main-loop efficiency: plain GEMMs (1x1 conv) with large K"""
    for (m, k, n) in ((32768, 4096, 1024), (32768, 1024, 1024), (32768, 256, 1024), (8192, 4096, 4096), (32768, 4096, 64), (32768, 4096, 32)):
        x = torch.randn(m, 1, 1, k, device="cuda")
        w = torch.randn(n, k, 1, 1, device="cuda")
        fl = 2.0 * m * k * n
        t = timeit(lambda: K.conv2d_fwd(x, w, None, 1, 0, 1), 5)
        dy = torch.randn(m, 1, 1, n, device="cuda")
        t2 = timeit(lambda: K.conv2d_wgrad(x, dy, w.shape, 1, 0, 1), 5)
        print(f"M={m} K={k} N={n}: fwd {t*1e6:9.1f} us {fl/t/1e12:7.2f} TF/s | wgrad {t2*1e6:9.1f} us {fl/t2/1e12:7.2f} TF/s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gemm":
    gemm_main()


def ksweep_main():
    """time = a + b*K for fixed (M, N): separates per-tile fixed cost (prologue / epilogue) from the main-loop rate"""
    for (m, n) in ((25088, 960), (25088, 128), (6272, 1536), (6272, 256), (1568, 256)):
        for k in (32, 64, 128, 256, 512, 1024, 2048):
            x = torch.randn(m, 1, 1, k, device="cuda")
            w = torch.randn(n, k, 1, 1, device="cuda")
            fl = 2.0 * m * k * n
            t = timeit(lambda: K.conv2d_fwd(x, w, None, 1, 0, 1), 10)
            print(f"M={m} N={n} K={k}: fwd {t*1e6:9.1f} us {fl/t/1e12:7.2f} TF/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ksweep":
    ksweep_main()
