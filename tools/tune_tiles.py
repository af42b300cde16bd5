"""Measure every tile of the GEMM menus on the step's conv shapes (pinned through suhpe_conv2d_set_tile):
    python tools/tune_tiles.py [batch ...]
prints, per (shape, pass), the time of each tile and the winner; the winners feed the selection tables in conv_igemm.hip.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
import kbench

lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p


def short(name):
    kind = "P" if "persistent" in name else ("H" if "3x3" in name else ("R" if "resident" in name else ""))
    a = name.split("<")[1].rstrip(">").split(",")
    return kind + "x".join(a[:2]) + ("w" + "x".join(a[2:4]) if len(a) > 3 else "") + ("v1" if len(a) > 4 and a[4] == "1" else "")


def menu(pass_id):
    return {i: short(lib.suhpe_conv2d_tile_name(pass_id, i).decode()) for i in range(lib.suhpe_conv2d_tile_count(pass_id))}
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")


def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e-3


batches = [int(v) for v in sys.argv[1:]] or [128, 32]
for batch in batches:
    for cin, cout, k, st, g, hw, cnt in kbench.effnet_convs():
        if g > 1 or cin == 3 or (os.environ.get("ONLY3") and (k != 3 or st != 1)):
            continue
        x = torch.randn(batch, hw, hw, cin, device="cuda")
        w = K.to_filter_layout(torch.randn(cout, cin, k, k, device="cuda"))
        pad = (k - 1) // 2
        d = K.conv_desc(x.shape, w.shape, st, pad, g)
        y = K.conv2d_fwd(x, w, None, st, pad, g)
        dy = torch.randn_like(y)
        fl = K.conv_flops(d)
        for pid, fn in ((0, lambda: K.conv2d_fwd(x, w, None, st, pad, g)),
                        (1, lambda: K.conv2d_dgrad(dy, w, x.shape, st, pad, g)),
                        (2, lambda: K.conv2d_wgrad(x, dy, w.shape, st, pad, g))):
            res = {}
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
            res["auto"] = timeit(fn)
            for idx, name in menu(pid).items():
                if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, idx) != 0:
                    continue
                res[name] = timeit(fn)
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
            best = min((v, k_) for k_, v in res.items() if k_ != "auto")
            line = " ".join(f"{k_}={v*1e6:.0f}" for k_, v in res.items())
            print(f"n={batch} {(cin, cout, k, st, hw)} pass{pid} x{cnt}: best {best[1]} {fl/best[0]/1e12:.1f}TF (auto {fl/res['auto']/1e12:.1f}) | {line}", flush=True)
