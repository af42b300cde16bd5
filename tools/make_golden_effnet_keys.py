"""State-dict key list and tensor shapes of torchvision 0.14.1's `efficientnet_v2_s` as the reference uses it (src/networks.py:301-332: the
backbone `features.*` of `models.efficientnet_v2_s`, `classifier` replaced by the 8-entry regression head), written WITHOUT instantiating
any module of this repo or of the oracle: the names follow torchvision's module tree (Conv2dNormActivation = `.0` conv / `.1` norm;
FusedMBConv / MBConv `.block.<i>`; SqueezeExcitation `.fc1` / `.fc2`), the sizes its published stage table.

    python tools/make_golden_effnet_keys.py        # -> tests/golden/effnetv2s_state_dict.json

torchvision is not installed in this image, so this is a second, independent restatement, not an import; what pins it from outside is the
parameter count torchvision publishes for the model: 21,458,488 with the 1000-class ImageNet classifier (checked below).  A released
reference checkpoint (`model_state_dict` of *-effinetv2-*.pth) has exactly these keys."""
import hashlib, json, os

STAGES = [("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4),
          ("mb", 4, 3, 2, 64, 128, 6), ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)]


def make_divisible(v, d=8):
    n = max(d, int(v + d / 2) // d * d)
    return n + d if n < 0.9 * v else n


def entries():
    out = []

    def conv(prefix, cout, cin_g, k):
        out.append((prefix + ".weight", [cout, cin_g, k, k]))

    def bn(prefix, c):
        for name, shape in (("weight", [c]), ("bias", [c]), ("running_mean", [c]), ("running_var", [c]), ("num_batches_tracked", [])):
            out.append((f"{prefix}.{name}", shape))

    def cna(prefix, cin, cout, k, groups=1):            # Conv2dNormActivation: bias-free conv + BatchNorm2d
        conv(prefix + ".0", cout, cin // groups, k)
        bn(prefix + ".1", cout)

    cna("features.0", 3, 24, 3)
    for s, (kind, e, k, stride, cin, cout, n) in enumerate(STAGES, start=1):
        for b in range(n):
            i = cin if b == 0 else cout
            exp = make_divisible(i * e)
            p = f"features.{s}.{b}.block"
            if kind == "fused":
                if exp != i:
                    cna(p + ".0", i, exp, k)
                    cna(p + ".1", exp, cout, 1)
                else:
                    cna(p + ".0", i, cout, k)
            else:
                j = 0
                if exp != i:
                    cna(f"{p}.{j}", i, exp, 1); j += 1
                cna(f"{p}.{j}", exp, exp, k, groups=exp); j += 1
                sq = max(1, i // 4)
                out.append((f"{p}.{j}.fc1.weight", [sq, exp, 1, 1])); out.append((f"{p}.{j}.fc1.bias", [sq]))
                out.append((f"{p}.{j}.fc2.weight", [exp, sq, 1, 1])); out.append((f"{p}.{j}.fc2.bias", [exp])); j += 1
                cna(f"{p}.{j}", exp, cout, 1)
    cna("features.7", 256, 1280, 1)
    backbone = list(out)
    # the reference's regression head (networks.py:309-319): Dropout, Linear, BatchNorm1d, ReLU6, Linear, BatchNorm1d, ReLU6, Linear
    for idx, (fi, fo) in ((1, (1280, 512)), (4, (512, 128)), (7, (128, 9))):
        out.append((f"classifier.{idx}.weight", [fo, fi])); out.append((f"classifier.{idx}.bias", [fo]))
        if idx != 7:
            bn(f"classifier.{idx + 1}", fo)
    out.sort(key=lambda kv: order_key(kv[0]))
    return backbone, out


def order_key(name):                                     # module-tree order: numeric components compare as numbers
    return [int(p) if p.isdigit() else p for p in name.replace("fc1", "a1").replace("fc2", "a2").split(".")[:-1]]


def numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


if __name__ == "__main__":
    backbone, full = entries()
    is_param = lambda k: not k.endswith(("running_mean", "running_var", "num_batches_tracked"))
    nb = sum(numel(s) for k, s in backbone if is_param(k))
    assert nb + 1280 * 1000 + 1000 == 21_458_488, nb      # torchvision's published num_params of efficientnet_v2_s (ImageNet classifier)
    nfull = sum(numel(s) for k, s in full if is_param(k))
    doc = {"source": "torchvision 0.14.1 efficientnet_v2_s module tree + the reference's classifier (src/networks.py:301-332); restated, see the script",
           "backbone_params": nb, "params_with_reference_head": nfull, "entries": [[k, s] for k, s in full]}
    doc["sha256"] = hashlib.sha256(json.dumps(doc["entries"]).encode()).hexdigest()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "effnetv2s_state_dict.json")
    json.dump(doc, open(path, "w"), indent=0)
    print(len(full), "entries;", nb, "backbone parameters;", nfull, "with the head ->", os.path.normpath(path))
