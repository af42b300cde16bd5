"""Re-measure the pointwise forward / input-gradient entries of the tile database whose OUTPUT has at most 64 channels against the narrow LDS-DMA
tile (gemm_nt_dma<256,64>, conv_dma.inl).  The forward is timed the way the training step runs it: with the BatchNorm statistics epilogue, and for
the FusedMBConv project convs (cin = 4 cout or 2 cout: 192 -> 48, 96 -> 48, 256 -> 64, 192 -> 64) with the producer's BatchNorm + SiLU as the
loader transform (XF = 1, two batch segments at n = 160); an entry moves when the new tile is at least 3 % faster (cold runs, minimum of 7).

    python tools/retune_dma_narrow.py gpurun_out/tiles_mi355x.json [batch sizes, default 128,160]
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
narrow = names.index("gemm_nt_dma<256,64>")
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
for key in sorted(db["tiles"]):
    pid, n, h, w, cin, cout, r, s_, stride, pad, groups = (int(v) for v in key.split(","))
    if pid > 1 or groups != 1 or r != 1 or s_ != 1 or stride != 1 or pad != 0 or n not in batches or db["tiles"][key] not in names:
        continue
    K_, N_ = (cin, cout) if pid == 0 else (cout, cin)
    if K_ % 32 or N_ > 64 or h < 14 or max(n * h * w * cin, n * h * w * cout) * 4 >= (1 << 31):
        continue
    torch.cuda.empty_cache()
    x = torch.randn(n, h, w, cin, device="cuda")
    wt = K.to_filter_layout(torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5)
    dy = torch.randn(n, h, w, cout, device="cuda")
    d = K.conv_desc(x.shape, wt.shape, 1, 0, 1)
    xf_form = pid == 0 and cin in (2 * cout, 3 * cout, 4 * cout)
    seg = 32 if n == 160 else None
    if pid == 0 and xf_form:
        nseg = 2 if seg else 1
        sc, sh = torch.rand(nseg, cin, device="cuda") + 0.5, torch.randn(nseg, cin, device="cuda") * 0.3
        run = lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, seg=seg, xf=(sc, sh, None, "silu", h * w, seg * h * w if seg else 0))
    elif pid == 0:
        run = lambda: K.conv2d_fwd_stats(x, wt, 1, 0, 1, seg=seg)
    else:
        run = lambda: K.conv2d_dgrad(dy, wt, x.shape, 1, 0, 1)
    cur = names.index(db["tiles"][key])
    times = {}
    for tile in (cur, narrow):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) != 0:
            continue
        times[tile] = cold_min(run)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    if narrow not in times:
        continue
    seen += 1
    nbytes = 4.0 * n * h * w * (cin + cout)
    line = (f"{key:42s} {'xf' if xf_form else ('st' if pid == 0 else 'dg')} {names[cur]:34s} {times[cur]*1e3:7.1f} us   narrow {times[narrow]*1e3:7.1f} us "
            f"({nbytes/times[narrow]/1e9:5.2f} TB/s of operands)")
    if cur != narrow and times[narrow] < 0.97 * times[cur]:
        db["tiles"][key] = names[narrow]
        moved += 1
        gain_us += (times[cur] - times[narrow]) * 1e3
        line += "   -> moved"
    print(line, flush=True)
print(f"{seen} narrow-output pointwise entries measured, {moved} moved to the narrow LDS-DMA tile, {gain_us:.0f} us saved over one launch of each")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(db, open(out, "w"), indent=0)
