"""Worker of tests/test_gpu_kernels.py::test_split_bf16_tiles (run with SUHPE_SPLIT_BF16=1: the switch is read when the library loads).
Every split-bf16 tile of the NT menu (igemm_nt_kernel<..., split>: an fp32 product as six bf16 MFMA planes) against a float64
reference on forward / input-gradient shapes of every loader mode (pointwise, 3x3 stride 1, 3x3 stride 2 and its parity-class
input gradient, grouped), next to the fp32-MFMA tile of the same size: the split result must be as close to the reference."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from semiuhpe_amd import _lib, kernels as K

lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
split = [i for i, n in enumerate(names) if "split" in n]
assert len(split) == 4, names
g = torch.Generator().manual_seed(5)
worst = 0.0
for (n, cin, cout, k, stride, pad, groups, h, w) in [(20, 160, 320, 1, 1, 0, 1, 14, 14), (6, 48, 192, 3, 1, 1, 1, 28, 28), (5, 24, 96, 3, 2, 1, 1, 30, 30),
                                                      (3, 64, 64, 3, 1, 1, 2, 14, 14), (70, 512, 36, 1, 1, 0, 1, 1, 1)]:
    x = (torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.3).cuda()
    wt = (torch.randn(cout, cin // groups, k, k, generator=g) / (k * (cin // groups) ** 0.5)).cuda()
    wg = K.to_filter_layout(wt)
    d = K.conv_desc(x.shape, wg.shape, stride, pad, groups)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = F.conv2d(x64, wt.double(), None, stride, pad, 1, groups)
    dy = torch.randn(y64.shape, generator=g, dtype=torch.float64).cuda().float()
    y64.backward(dy.double())
    y_ref, dx_ref = y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1)
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous()
    mag_y = F.conv2d(x.double().abs().permute(0, 3, 1, 2), wt.double().abs(), None, stride, pad, 1, groups).permute(0, 2, 3, 1)     # sum |a||b| per output
    for pid, run, ref, mag in ((0, lambda: K.conv2d_fwd(x, wg, None, stride, pad, groups), y_ref, mag_y),
                               (1, lambda: K.conv2d_dgrad(dy_nhwc, wg, x.shape, stride, pad, groups), dx_ref, None)):
        scale = mag if mag is not None else ref.abs().mean() * 30
        errs = {}
        for tile in [1] + split:                        # tile 1: the fp32 64x64 tile
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile) == 0, (names[tile], K.last_error() if hasattr(K, "last_error") else "")
            out = run().double()
            errs[tile] = float(((out - ref).abs() / scale).max())
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, -1)
        base = errs.pop(1)
        for tile, e in errs.items():
            assert e <= 1.5 * base + 1e-8, (names[tile], pid, e, base)
            worst = max(worst, e / base)
    # the statistics epilogue and the folded-affine epilogue ride on the same accumulators
    if groups == 1 and k == 1:
        for tile in split:
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, tile) == 0
            res = K.conv2d_fwd_stats(x, wg, stride, pad, groups)
            rows = res[0].numel() // cout
            mean, rstd = K.bn_stats_from_partials(res[1], res[2], rows, cout, 1e-3, 0.1)
            m0 = y_ref.reshape(-1, cout).mean(0)
            assert float((mean.double() - m0).abs().max()) < 1e-5, names[tile]
            sc, sh = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda()
            ya = K.conv2d_fwd_affine(x, wg, sc, sh, "silu", None, False, stride, pad, groups)
            assert float((ya.double() - F.silu(y_ref * sc.double() + sh.double())).abs().max()) < 2e-5, names[tile]
            lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1)
print("SPLIT_TILES_OK worst error ratio vs the fp32 tile %.2f" % worst)

# ---- network level: EfficientNetV2-S with EVERY forward / input-gradient GEMM pinned to a split tile, against the CPU oracle at the
# tolerances of tests/test_gpu_effnet.py (rotation matrices 1e-4, outputs 2e-4, gradients 2e-3 of each tensor's max)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from parity_utils import MaskTape, grad_mismatches, rel_err
from oracle import networks as ON, fisher as OF
from semiuhpe_amd import layers as L
from semiuhpe_amd.src.efficientnet_v2 import EfficientNetV2S

pinned = set()


def pin_split(d, pass_id, run):
    if pass_id < 2 and not (d.cin % 4 or d.cout % 4 or (d.cin // d.groups) % 4 or (d.cout // d.groups) % 4):
        key = (pass_id, d.n, d.h, d.w, d.cin, d.cout, d.r, d.s, d.stride, d.pad, d.groups)
        if key not in pinned:
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pass_id, split[0]) == 0          # 64x64 split tile
            pinned.add(key)


K._autotune, K.AUTOTUNE = pin_split, True
torch.manual_seed(0)
ref = ON.EfficientNetV2S()
gg = torch.Generator().manual_seed(1)
for mod in ref.modules():
    if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
        mod.running_mean.copy_(torch.randn(mod.num_features, generator=gg) * 0.1)
        mod.running_var.copy_(torch.rand(mod.num_features, generator=gg) * 0.5 + 0.75)
        mod.weight.data.copy_(torch.rand(mod.num_features, generator=gg) * 0.5 + 0.75)
        mod.bias.data.copy_(torch.randn(mod.num_features, generator=gg) * 0.1)
for mod in ref.classifier:
    if isinstance(mod, torch.nn.Linear):
        mod.weight.data.mul_(20.0)
hip = EfficientNetV2S()
hip.load_state_dict(ref.state_dict())
hip.cuda()
ref.eval(); hip.eval()
xe = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(2))
with torch.no_grad():
    ye_ref, ye = ref(xe), hip(xe.cuda())
assert rel_err(ye, ye_ref) < 2e-4
assert float((K.fisher_svd_project(ye).cpu() - OF.batch_torch_A_to_R(ye_ref)).abs().max()) < 1e-4
ref.train(); hip.train()
tape = MaskTape(7)
xt = torch.randn(6, 3, 96, 96, generator=torch.Generator().manual_seed(3))
dyt = torch.randn(6, 9, generator=torch.Generator().manual_seed(4))
bn_before = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k}
ON.set_mask_hook(tape.record)
yt_ref = ref(xt)
yt_ref.backward(dyt)
ON.set_mask_hook(None)
hip.load_state_dict({**hip.state_dict(), **bn_before})
L.set_mask_hook(tape.replay)
yt = hip(xt.cuda())
yt.backward(dyt.cuda())
L.set_mask_hook(None)
assert tape.done() and rel_err(yt, yt_ref) < 2e-4
bad = grad_mismatches({n: p.grad for n, p in ref.named_parameters()}, {n: p.grad for n, p in hip.named_parameters()}, 2e-3)
assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:6]}"
assert len(pinned) > 50, len(pinned)
print("SPLIT_NETWORK_OK %d (shape, pass) pairs on split tiles" % len(pinned))
