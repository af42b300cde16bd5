"""Algorithmic work of the hot path: per-layer multiply-accumulates and conv/linear activation traffic of one
image-forward of the three backbones, and from them the FLOP / byte totals and the roofline time T_roof of one SSL step.

    python tools/count_work.py                    # per-network totals + the C2 / C3 / C5 step table (runs on CPU, no GPU needed)
    python tools/count_work.py effinetv2 -v       # + the per-layer table

Conventions (§8d): MAC = one multiply-accumulate of a dense / grouped / depthwise conv, Linear, or attention matmul; FLOP = 2 MAC;
"activation elems" = inputs + outputs of every conv / linear (what an ideally fused implementation must move, fp32);
backward = 2x forward; an SSL step with B labeled and muB unlabeled images runs B + 2 muB image-forwards (student labeled, teacher
weak, student strong) and B + muB image-backwards.   T_roof = sum over layers of max(FLOP / P_mfma, bytes / BW_hbm).
The layer lists are restated from the networks' structure (src/networks.py, src/repvgg.py, src/tiny_vit.py of the reference,
torchvision 0.14.1 efficientnet_v2_s) and are cross-checked below against the parameter counts of this repo's modules.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

P_MFMA = 157.3e12          # fp32 dense MFMA peak, FLOP/s (MI355X_MICROARCH.md)
BW_HBM = 6.3e12            # achievable HBM rate SURVEY.md §8d prices the streaming layers at (8.0 TB/s spec)


class Work:
    def __init__(self):
        self.layers = []                      # (name, kind, macs, in_elems, out_elems, params)

    def conv(self, name, h, cin, cout, k, stride=1, groups=1, bias=False):
        ho = (h + 2 * (k // 2) - k) // stride + 1
        macs = ho * ho * cout * (cin // groups) * k * k
        kind = "dw" if groups == cin == cout and groups > 1 else "conv"
        self.layers.append((name, kind, macs, h * h * cin, ho * ho * cout, cout * (cin // groups) * k * k + (cout if bias else 0)))
        return ho

    def linear(self, name, tokens, cin, cout, bias=True):
        self.layers.append((name, "linear", tokens * cin * cout, tokens * cin, tokens * cout, cin * cout + (cout if bias else 0)))

    def attn(self, name, windows, heads, n, d):
        # q k^T and p v: 2 * n*n*d MACs per head per window; traffic: q, k, v in, o out
        self.layers.append((name, "attention", windows * heads * 2 * n * n * d, windows * heads * 3 * n * d, windows * heads * n * d, 0))

    def extra_params(self, name, count):
        self.layers.append((name, "param", 0, 0, 0, count))

    def totals(self):
        t = {"macs": 0, "dense_macs": 0, "dw_macs": 0, "attn_macs": 0, "elems": 0, "conv_elems": 0, "dw_elems": 0, "params": 0, "convs": 0}
        for name, kind, macs, ein, eout, params in self.layers:
            t["macs"] += macs
            t["params"] += params
            if kind == "attention":
                t["attn_macs"] += macs
            elif kind == "dw":
                t["dw_macs"] += macs
                t["dw_elems"] += ein + eout
            else:
                t["dense_macs"] += macs
            if kind in ("conv", "dw"):
                t["convs"] += 1
                t["conv_elems"] += ein + eout
            if kind in ("conv", "dw", "linear"):
                t["elems"] += ein + eout
        return t

    def t_roof(self):
        """seconds for one image-forward at the roofline: each layer at max(flop time, byte time)"""
        return sum(max(2 * macs / P_MFMA, 4 * (ein + eout) / BW_HBM) for _, kind, macs, ein, eout, _ in self.layers if kind != "param")


def bn(w, name, c):
    w.extra_params(name, 2 * c)


def head(w, cin, h1, h2, nout):
    """Dropout -> Linear -> BN1d -> ReLU6 -> Linear -> BN1d -> ReLU6 -> Linear (src/networks.py:309-319 and siblings)"""
    w.linear("head.fc1", 1, cin, h1); bn(w, "head.bn1", h1)
    w.linear("head.fc2", 1, h1, h2); bn(w, "head.bn2", h2)
    w.linear("head.fc3", 1, h2, nout)


def effinetv2_s():
    """torchvision 0.14.1 efficientnet_v2_s + the reference's regression head (src/networks.py:301-332)"""
    w = Work()
    h = w.conv("features.0", 224, 3, 24, 3, 2); bn(w, "features.0.bn", 24)
    cfg = [("fused", 1, 1, 24, 24, 2), ("fused", 4, 2, 24, 48, 4), ("fused", 4, 2, 48, 64, 4),
           ("mb", 4, 2, 64, 128, 6), ("mb", 6, 1, 128, 160, 9), ("mb", 6, 2, 160, 256, 15)]
    for si, (kind, e, stride, cin, cout, reps) in enumerate(cfg, start=1):
        for b in range(reps):
            ci, s = (cin, stride) if b == 0 else (cout, 1)
            n = f"features.{si}.{b}"
            if kind == "fused":
                if e == 1:
                    h = w.conv(n + ".conv3x3", h, ci, cout, 3, s); bn(w, n + ".bn", cout)
                else:
                    h = w.conv(n + ".expand3x3", h, ci, ci * e, 3, s); bn(w, n + ".bn0", ci * e)
                    w.conv(n + ".project1x1", h, ci * e, cout, 1); bn(w, n + ".bn1", cout)
            else:
                w.conv(n + ".expand1x1", h, ci, ci * e, 1); bn(w, n + ".bn0", ci * e)
                h = w.conv(n + ".dw3x3", h, ci * e, ci * e, 3, s, groups=ci * e); bn(w, n + ".bn1", ci * e)
                sq = max(1, ci // 4)
                w.linear(n + ".se.fc1", 1, ci * e, sq); w.linear(n + ".se.fc2", 1, sq, ci * e)
                w.conv(n + ".project1x1", h, ci * e, cout, 1); bn(w, n + ".bn2", cout)
    w.conv("features.7", h, 256, 1280, 1); bn(w, "features.7.bn", 1280)
    head(w, 1280, 512, 128, 9)
    return w


def repvgg_b1g2():
    """RepVGG-B1g2 in training form (src/repvgg.py:67-76,190-199; head src/networks.py:229-290)"""
    w = Work()
    widths, blocks = [64, 128, 256, 512, 2048], [1, 4, 6, 16, 1]
    h, cin, idx = 224, 3, 0
    for stage, (cout, reps) in enumerate(zip(widths, blocks)):
        for b in range(reps):
            stride = 2 if b == 0 else 1
            groups = 2 if (idx in range(2, 27, 2)) else 1
            n = f"layer{stage}.{b}" if stage else "layer0"
            w.conv(n + ".rbr_1x1", h, cin, cout, 1, stride, groups); bn(w, n + ".rbr_1x1.bn", cout)
            if cin == cout and stride == 1:
                bn(w, n + ".rbr_identity", cout)
            h = w.conv(n + ".rbr_dense", h, cin, cout, 3, stride, groups); bn(w, n + ".rbr_dense.bn", cout)
            cin = cout
            idx += 1
    head(w, 2048, 512, 128, 9)
    return w


def tinyvit_21m():
    """tiny_vit_21m_224 (src/tiny_vit.py:453-591; head src/networks.py:419-428)"""
    w = Work()
    dims, depths, heads, ws = (96, 192, 384, 576), (2, 2, 6, 2), (3, 6, 12, 18), (7, 7, 14, 7)
    h = w.conv("patch_embed.0", 224, 3, 48, 3, 2); bn(w, "patch_embed.0.bn", 48)
    h = w.conv("patch_embed.2", h, 48, 96, 3, 2); bn(w, "patch_embed.2.bn", 96)
    for i in range(4):
        d = dims[i]
        for b in range(depths[i]):
            n = f"layers.{i}.blocks.{b}"
            if i == 0:
                w.conv(n + ".conv1", h, d, 4 * d, 1); bn(w, n + ".bn1", 4 * d)
                w.conv(n + ".conv2", h, 4 * d, 4 * d, 3, 1, groups=4 * d); bn(w, n + ".bn2", 4 * d)
                w.conv(n + ".conv3", h, 4 * d, d, 1); bn(w, n + ".bn3", d)
            else:
                tokens, win = h * h, ws[i] * ws[i]
                w.extra_params(n + ".attn.norm", 2 * d)
                w.linear(n + ".attn.qkv", tokens, d, 3 * d)
                w.attn(n + ".attn", tokens // win, heads[i], win, d // heads[i])
                w.extra_params(n + ".attn.attention_biases", heads[i] * win)      # one bias per |offset|: ws*ws distinct offsets
                w.linear(n + ".attn.proj", tokens, d, d)
                w.conv(n + ".local_conv", h, d, d, 3, 1, groups=d); bn(w, n + ".local_conv.bn", d)
                w.extra_params(n + ".mlp.norm", 2 * d)
                w.linear(n + ".mlp.fc1", tokens, d, 4 * d); w.linear(n + ".mlp.fc2", tokens, 4 * d, d)
        if i < 3:
            o, n = dims[i + 1], f"layers.{i}.downsample"
            w.conv(n + ".conv1", h, d, o, 1); bn(w, n + ".bn1", o)
            h = w.conv(n + ".conv2", h, o, o, 3, 2, groups=o); bn(w, n + ".bn2", o)
            w.conv(n + ".conv3", h, o, o, 1); bn(w, n + ".bn3", o)
    w.extra_params("norm_head", 2 * 576)
    head(w, 576, 128, 64, 9)
    return w


def resnet50():
    """torchvision resnet50 (Bottleneck v1.5) + the reference's regression head (src/networks.py:443-500)"""
    w = Work()
    h = w.conv("conv1", 224, 3, 64, 7, 2); bn(w, "bn1", 64)
    h = (h + 2 - 3) // 2 + 1                                   # maxpool 3x3 s2 p1
    cin = 64
    for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
        for b in range(blocks):
            s, n = (stride if b == 0 else 1), f"layer{i}.{b}"
            w.conv(n + ".conv1", h, cin, planes, 1); bn(w, n + ".bn1", planes)
            h2 = w.conv(n + ".conv2", h, planes, planes, 3, s); bn(w, n + ".bn2", planes)
            w.conv(n + ".conv3", h2, planes, planes * 4, 1); bn(w, n + ".bn3", planes * 4)
            if s != 1 or cin != planes * 4:
                w.conv(n + ".downsample", h, cin, planes * 4, 1, s); bn(w, n + ".downsample.bn", planes * 4)
            h, cin = h2, planes * 4
    head(w, 2048, 512, 128, 9)
    return w


NETS = {"effinetv2": effinetv2_s, "repvgg": repvgg_b1g2, "tinyvit": tinyvit_21m, "resnet50": resnet50}
# per-GPU step shapes of the BASELINE configs: (network, B labeled, muB unlabeled)
CONFIGS = {"C1": ("effinetv2", 32, 0), "C2": ("effinetv2", 32, 128), "C3": ("repvgg", 64, 256), "C5": ("tinyvit", 32, 128)}


def step_work(net, B, muB):
    """FLOP, ideal-fusion bytes and roofline seconds of one training step (forward images B + 2 muB, backward images B + muB,
    backward = 2x forward); plus Adam (4 streams) and EMA (3 streams) over the parameters."""
    w = NETS[net]()
    t = w.totals()
    fwd, bwd = B + 2 * muB, B + muB
    units = fwd + 2 * bwd
    flop = units * 2 * t["macs"]
    nbytes = units * 4 * t["elems"] + 7 * 4 * t["params"]
    t_roof = units * w.t_roof() + 7 * 4 * t["params"] / BW_HBM
    return {"flop": flop, "bytes": nbytes, "t_flop": flop / P_MFMA, "t_hbm": nbytes / BW_HBM, "t_roof": t_roof,
            "flop_per_counted_image": flop / max(1, B + muB)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("net", nargs="?", choices=sorted(NETS))
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    for name in ([args.net] if args.net else sorted(NETS)):
        w = NETS[name]()
        t = w.totals()
        if args.verbose:
            for lname, kind, macs, ein, eout, params in w.layers:
                if kind != "param":
                    print(f"  {lname:34s} {kind:9s} {macs / 1e6:10.3f} MMAC  in {ein:9d}  out {eout:9d}  params {params}")
        print(f"{name}: {t['macs'] / 1e9:.3f} GMAC = {2 * t['macs'] / 1e9:.2f} GFLOP fwd/img (dense+linear {t['dense_macs'] / 1e9:.3f}, depthwise "
              f"{t['dw_macs'] / 1e9:.3f}, attention {t['attn_macs'] / 1e9:.3f}); {t['convs']} convs; conv activations in+out {t['conv_elems'] / 1e6:.1f} M elems "
              f"({4 * t['conv_elems'] / 1e6:.1f} MB fp32, depthwise {t['dw_elems'] / 1e6:.1f} M; with linear layers {t['elems'] / 1e6:.1f} M); params {t['params']:,}; "
              f"T_roof {w.t_roof() * 1e6:.1f} us/img-fwd")
    print()
    for cname, (net, B, muB) in CONFIGS.items():
        s = step_work(net, B, muB)
        print(f"{cname} {net} {B}+{muB}: {s['flop'] / 1e12:.2f} TFLOP/step, {s['bytes'] / 1e9:.1f} GB/step -> T_flop {s['t_flop'] * 1e3:.1f} ms, "
              f"T_hbm {s['t_hbm'] * 1e3:.1f} ms, T_roof {s['t_roof'] * 1e3:.1f} ms -> ceiling {(B + muB) / s['t_roof']:.0f} img/s; "
              f"{s['flop_per_counted_image'] / 1e9:.1f} GFLOP per counted image")


if __name__ == "__main__":
    main()
