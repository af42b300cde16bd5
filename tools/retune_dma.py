"""Re-measure the pointwise (1x1 / stride 1, one group, reduction % 32 == 0) forward and input-gradient entries of the tile database against the
LDS-DMA tile (gemm_nt_dma, conv_dma.inl, round 5) on synthetic operands of each entry's shape.  The forward is timed WITH its BatchNorm statistics
epilogue (how the training step runs it); an entry moves when the new tile is at least 3 % faster than the entry's current winner (cold runs: a
256 MiB fill before every launch, minimum of 7).

    python tools/retune_dma.py gpurun_out/tiles_mi355x.json [batch sizes, default 128,160]     (needs an MI355X; copy the result over semiuhpe_amd/tiles_mi355x.json)
"""
import ctypes, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from semiuhpe_amd import _lib, kernels as K  # noqa: E402

out = os.path.abspath(sys.argv[1])
batches = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128, 160]
src = os.environ.get("SUHPE_TILE_DB", os.path.join(ROOT, "semiuhpe_amd", "tiles_mi355x.json"))
db = json.load(open(src))
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
dma = names.index("gemm_nt_dma<128,128>")
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")


def cold_min(fn, reps=7):
    fn()
    best = float("inf")
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


moved, seen, gain_us = 0, 0, 0.0
cache_key, cache = None, None
for key in sorted(db["tiles"]):
    pid, n, h, w, cin, cout, r, s_, stride, pad, groups = (int(v) for v in key.split(","))
    if pid > 1 or groups != 1 or r != 1 or s_ != 1 or stride != 1 or pad != 0 or n not in batches or db["tiles"][key] not in names:
        continue
    if (cin if pid == 0 else cout) % 32 or max(n * h * w * cin, n * h * w * cout) * 4 >= (1 << 31):
        continue
    shape = (n, h, w, cin, cout)
    if shape != cache_key:
        cache = None
        torch.cuda.empty_cache()
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = K.to_filter_layout(torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5)
        dy = torch.randn(n, h, w, cout, device="cuda")
        cache_key, cache = shape, (x, wt, dy)
    x, wt, dy = cache
    d = K.conv_desc(x.shape, wt.shape, 1, 0, 1)
    run = {0: lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1), 1: lambda: K.conv2d_dgrad(dy, wt, x.shape, 1, 0, 1)}[pid]
    cur = names.index(db["tiles"][key])
    times = {}
    for tile in (cur, dma):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) != 0:
            continue
        times[tile] = cold_min(run)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    if dma not in times:
        continue
    seen += 1
    fl = 2.0 * n * h * w * cin * cout
    line = f"{key:44s} {names[cur]:34s} {times[cur]*1e3:7.1f} us ({fl/times[cur]/1e9:5.0f} TF/s)   dma {times[dma]*1e3:7.1f} us ({fl/times[dma]/1e9:5.0f} TF/s)"
    if cur != dma and times[dma] < 0.97 * times[cur]:
        db["tiles"][key] = names[dma]
        moved += 1
        gain_us += (times[cur] - times[dma]) * 1e3
        line += "   -> moved"
    print(line, flush=True)
print(f"{seen} eligible pointwise entries measured, {moved} moved to the LDS-DMA tile, {gain_us:.0f} us saved over one launch of each")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(db, open(out, "w"), indent=0)
