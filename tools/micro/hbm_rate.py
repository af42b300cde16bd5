"""Reference streaming rates on this device (torch built-ins, no custom kernel): copy / fill / sum over buffers well beyond
the 256 MiB Infinity Cache, and over a 92 MiB buffer (typical activation) cold and warm."""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (2048, 92):
    x = torch.randn(mb * 2**20 // 4, device="cuda"); y = torch.empty_like(x)
    b = x.numel() * 4
    print(f"{mb} MiB: copy {2*b/t(lambda: y.copy_(x))/1e9:6.0f} GB/s (r+w)  fill {b/t(lambda: y.zero_())/1e9:6.0f} GB/s  sum {b/t(lambda: x.sum())/1e9:6.0f} GB/s  "
          f"add {3*b/t(lambda: torch.add(x, y, out=y))/1e9:6.0f} GB/s", flush=True)
flush = torch.empty(1 << 28, device="cuda")
x = torch.randn(92 * 2**20 // 4, device="cuda"); y = torch.empty_like(x)
def cold(fn):
    tot = 0
    for _ in range(6):
        flush.zero_(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / 6 * 1e-3
b = x.numel() * 4
print(f"92 MiB cold: copy {2*b/cold(lambda: y.copy_(x))/1e9:6.0f} GB/s  sum {b/cold(lambda: x.sum())/1e9:6.0f} GB/s")
