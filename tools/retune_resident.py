"""Re-measure the pointwise (1x1 / stride 1) forward and input-gradient entries of the tile database against the operand-resident
tiles (gemm_nt_resident, conv_resident.inl), and every filter-gradient entry against the 128x32 tile that joined the TN menu, on
synthetic operands of each entry's shape; an entry moves to a new tile when that tile is at least 3 % faster than the entry's
current winner (cold runs: a 256 MiB fill before every launch, minimum of 5).

    python tools/retune_resident.py gpurun_out/tiles_mi355x.json     (needs an MI355X; copy the result over semiuhpe_amd/tiles_mi355x.json)
"""
import ctypes, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from semiuhpe_amd import _lib, kernels as K  # noqa: E402

out = os.path.abspath(sys.argv[1])
src = os.path.join(ROOT, "semiuhpe_amd", "tiles_mi355x.json")
db = json.load(open(src))
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names_nt = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
names_tn = [lib.suhpe_conv2d_tile_name(2, i).decode() for i in range(lib.suhpe_conv2d_tile_count(2))]
resident = [i for i, n in enumerate(names_nt) if "resident" in n]
new_tn = [i for i, n in enumerate(names_tn) if "<128,32," in n]
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")


def cold_min(fn, reps=5):
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
    names = names_tn if pid == 2 else names_nt
    if db["tiles"][key] not in names or (groups > 1 and groups == cin):
        continue
    if pid < 2 and (r != 1 or s_ != 1 or stride != 1 or pad != 0):
        continue
    if pid == 2 and (r != 1 or stride != 1):           # (pointwise filter gradients: where Cin = 160 / 96 / 48 pads a 64- or 128-wide tile)
        continue
    if max(n * h * w * cin, n * h * w * cout) * 4 > (6 << 30):
        continue
    shape = (n, h, w, cin, cout, groups)
    if shape != cache_key:
        cache = None
        torch.cuda.empty_cache()
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = K.to_filter_layout(torch.randn(cout, cin // groups, 1, 1, device="cuda") / (cin // groups) ** 0.5)
        dy = torch.randn(n, h, w, cout, device="cuda")
        cache_key, cache = shape, (x, wt, dy)
    x, wt, dy = cache
    d = K.conv_desc(x.shape, wt.shape, 1, 0, groups)
    run = {0: lambda: K.conv2d_fwd(x, wt, None, 1, 0, groups), 1: lambda: K.conv2d_dgrad(dy, wt, x.shape, 1, 0, groups),
           2: lambda: K.conv2d_wgrad(x, dy, wt.shape, 1, 0, groups)}[pid]
    cur = names.index(db["tiles"][key])
    times = {}
    for tile in [cur] + (new_tn if pid == 2 else resident):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) != 0:
            continue
        times[tile] = cold_min(run)
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
    seen += 1
    best = min(times, key=times.get)
    fl = 2.0 * n * h * w * (cin // groups) * cout
    line = f"{key:44s} " + "  ".join(f"{names[t].replace('igemm_nt_kernel', 'nt').replace('igemm_tn_kernel', 'tn').replace('gemm_nt_persistent', 'P').replace('gemm_nt_resident', 'R')}={times[t]*1e3:.1f}us" for t in times)
    if best != cur and times[best] < 0.97 * times[cur]:
        db["tiles"][key] = names[best]
        moved += 1
        gain_us += (times[cur] - times[best]) * 1e3
        line += f"   -> {names[best]} ({fl/times[best]/1e9:.0f} TF/s)"
    print(line, flush=True)
print(f"{seen} pointwise entries measured, {moved} moved to a new tile, {gain_us:.0f} us saved over one launch of each")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({"device": db["device"], "key": db["key"], "tiles": dict(sorted(db["tiles"].items()))}, open(out, "w"), indent=0)
