"""Window attention kernels of the C5 step in isolation (TinyViT-21M, 32 + 128 images): the probability-keeping kernels of round 3 against the
recomputing ones of round 5, forward (training / inference) and backward, cold-ish caches (a 512 MB sweep between launches).
    python tools/attnbench.py [B [window]]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K
from semiuhpe_amd.src import tiny_vit as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
flush = torch.empty(128 << 20, device="cuda")


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


ONLY = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for ws, heads, H in ((7, 6, 28), (14, 12, 14), (7, 18, 7)):
    if ONLY and (ws != ONLY or (ONLY == 7 and H != 28)):
        continue
    att = T.Attention(heads * 32, 32, heads, ws).cuda()
    att.attention_biases.data.normal_(0, 0.5)
    qkv = torch.randn(B * H * H, heads * 96, device="cuda")
    dout = torch.randn(B * H * H, heads * 32, device="cuda")
    idx, noff, inv = att._index(qkv.device), att.attention_biases.shape[1], att._inverse_index(qkv.device)
    tb = att.attention_biases
    out, probs = K.window_attn_fwd(qkv, tb, idx, B, H, H, ws, heads, att.scale)
    out2, stats = K.window_attn_fwd_rs(qkv, tb, idx, B, H, H, ws, heads, att.scale)
    db = torch.zeros(heads, noff, device="cuda")
    n = ws * ws
    flop_f = 2 * 2 * n * n * 32 * (B * (H // ws) ** 2 * heads)
    rows = [
        ("fwd  keep-P  (train)", lambda: K.window_attn_fwd(qkv, tb, idx, B, H, H, ws, heads, att.scale), flop_f),
        ("fwd  keep-P  (infer)", lambda: K.window_attn_fwd(qkv, tb, idx, B, H, H, ws, heads, att.scale, save_probs=False), flop_f),
        ("fwd  recompute (train)", lambda: K.window_attn_fwd_rs(qkv, tb, idx, B, H, H, ws, heads, att.scale), flop_f),
        ("fwd  recompute (infer)", lambda: K.window_attn_fwd_rs(qkv, tb, idx, B, H, H, ws, heads, att.scale, save_stats=False), flop_f),
        ("bwd  keep-P", lambda: K.window_attn_bwd(qkv, probs, dout, idx, B, H, H, ws, heads, noff, att.scale, db, inverse=inv), flop_f * 2.5),
        ("bwd  recompute", lambda: K.window_attn_bwd_rs(qkv, out2, stats, dout, tb, idx, B, H, H, ws, heads, noff, att.scale, db, inverse=inv), flop_f * 2.5),
    ]
    print(f"window {ws}x{ws}, {heads} heads, {H}x{H} map, {B} images ({flop_f/1e9:.2f} GFLOP forward)")
    for name, fn, fl in rows:
        us = timed(fn)
        print(f"   {name:24s} {us:8.1f} us   {fl/us/1e6:6.1f} TF/s (useful)")
