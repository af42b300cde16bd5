"""Golden vectors for RepVGG-B1g2 and TinyViT-21M by IMPORTING the reference modules (read-only /root/reference).

    python tools/make_golden_backbones.py        # -> tests/golden/backbones.npz  (+ block-level fixtures)

State dicts are far too large to commit (180 MB), so weights come from a deterministic recipe
(`tests/parity_utils.seeded_state_dict`: seeded generator walked over the state_dict in key order) that the tests
re-run; the fixture stores the inputs and the reference's outputs only:
  * eval-mode network outputs (2,9) on 64x64 (RepVGG) / 224x224 (TinyViT, its window geometry is fixed) inputs,
  * train-mode outputs with Dropout / DropPath disabled (p set to 0 on the instances), gradients of a few named
    parameters after `out.backward(dy)`, and the BatchNorm running stats after that step,
  * full tensors for one RepVGG block, one TinyViT block, one Attention, one MBConv, one PatchMerging.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import ref_shims  # noqa: E402

ref_shims.install()
from parity_utils import seeded_state_dict  # noqa: E402

import io, contextlib  # noqa: E402,E401
with contextlib.redirect_stdout(io.StringIO()):          # RepVGG prints one line per block
    from src.networks import RepVggNet, get_TinyViT  # noqa: E402  (reference)
    from src import repvgg as ref_repvgg, tiny_vit as ref_tv  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
GRAD_KEYS_REPVGG = ["layer0.rbr_dense.conv.weight", "layer1.1.rbr_1x1.conv.weight", "layer2.3.rbr_identity.weight",
                    "layer3.7.rbr_dense.bn.bias", "layer4.0.rbr_dense.conv.weight", "linear_reg.1.weight", "linear_reg.7.bias"]
GRAD_KEYS_TINYVIT = ["patch_embed.seq.0.c.weight", "layers.0.blocks.1.conv2.c.weight", "layers.1.blocks.0.attn.attention_biases",
                     "layers.1.blocks.0.attn.qkv.weight", "layers.2.blocks.3.mlp.fc1.weight", "layers.2.downsample.conv2.c.weight",
                     "layers.3.blocks.1.attn.norm.weight", "layers.3.blocks.1.local_conv.bn.weight", "norm_head.weight",
                     "head.1.weight", "head.7.bias"]


def put(store, key, arr, limit=20000):
    """Large tensors are stored as a strided sample of <= 4096 values plus sum and abs-sum (float64)."""
    arr = np.asarray(arr)
    if arr.size <= limit:
        store[key] = arr
        return
    flat = arr.reshape(-1)
    stride = max(1, flat.size // 4096)
    store[key + "@sample"] = flat[::stride][:4096].copy()
    store[key + "@stride"] = np.int64(stride)
    store[key + "@sums"] = np.array([flat.astype(np.float64).sum(), np.abs(flat.astype(np.float64)).sum()])


def disable_random_drops(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0


def run_net(model, x, dy, grad_keys, store, tag):
    model.eval()
    with torch.no_grad():
        store[tag + "/eval_out"] = model(x).numpy()
    model.train()
    disable_random_drops(model)
    for p in model.parameters():
        p.grad = None
    out = model(x)
    out.backward(dy)
    store[tag + "/train_out"] = out.detach().numpy()
    params = dict(model.named_parameters())
    for k in grad_keys:
        put(store, tag + "/grad/" + k, params[k].grad.numpy().copy())
    sd = model.state_dict()
    bn_keys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    store[tag + "/bn_keys"] = np.array(bn_keys[:6] + bn_keys[-6:])
    for k in bn_keys[:6] + bn_keys[-6:]:
        store[tag + "/bn/" + k] = sd[k].numpy().copy()


def main():
    torch.set_num_threads(8)
    store = {}
    g = torch.Generator().manual_seed(0)
    # ---------------- RepVGG-B1g2
    with contextlib.redirect_stdout(io.StringIO()):
        rep = RepVggNet(pretrained=False, num_classes=9)
    rep.load_state_dict(seeded_state_dict(rep.state_dict(), 11))
    x = torch.randn(4, 3, 64, 64, generator=g)
    dy = torch.randn(4, 9, generator=g)
    store["repvgg/x"], store["repvgg/dy"] = x.numpy(), dy.numpy()
    store["repvgg/num_state_entries"] = np.int64(len(rep.state_dict()))
    run_net(rep, x, dy, GRAD_KEYS_REPVGG, store, "repvgg")
    # one block with full tensors: 64 -> 128, stride 2 and 128 -> 128 groups 2 (identity branch present)
    for tag, (cin, cout, stride, groups) in {"blockA": (64, 128, 2, 1), "blockB": (128, 128, 1, 2)}.items():
        with contextlib.redirect_stdout(io.StringIO()):
            blk = ref_repvgg.RepVGGBlock(cin, cout, 3, stride=stride, padding=1, groups=groups)
        blk.load_state_dict(seeded_state_dict(blk.state_dict(), 5))
        xb = torch.randn(3, cin, 12, 12, generator=g).requires_grad_(True)
        blk.train()
        yb = blk(xb)
        dyb = torch.randn(yb.shape, generator=g)
        yb.backward(dyb)
        store[f"repvgg_{tag}/x"], store[f"repvgg_{tag}/dy"] = xb.detach().numpy(), dyb.numpy()
        store[f"repvgg_{tag}/y"], store[f"repvgg_{tag}/dx"] = yb.detach().numpy(), xb.grad.numpy()
        for k, p in blk.named_parameters():
            put(store, f"repvgg_{tag}/grad/{k}", p.grad.numpy())
    # ---------------- TinyViT-21M
    cfg = SimpleNamespace(num_classes=9)
    tv = get_TinyViT(cfg, pretrain=False, model_name="21m")
    tv.load_state_dict(seeded_state_dict(tv.state_dict(), 13))
    x = torch.randn(2, 3, 224, 224, generator=g) * 0.5
    dy = torch.randn(2, 9, generator=g)
    store["tinyvit/x_seed"] = np.int64(123)          # the 224x224 input is regenerated from this seed in the test
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(123)) * 0.5
    store["tinyvit/dy"] = dy.numpy()
    store["tinyvit/num_state_entries"] = np.int64(len(tv.state_dict()))
    store["tinyvit/bias_idx_7"] = tv.layers[1].blocks[0].attn.attention_bias_idxs.numpy()
    store["tinyvit/bias_idx_14_row0"] = tv.layers[2].blocks[0].attn.attention_bias_idxs.numpy()[:2]
    run_net(tv, x, dy, GRAD_KEYS_TINYVIT, store, "tinyvit")
    # block-level: TinyViTBlock(192, 28x28, 6 heads, ws 7), Attention alone, MBConv, PatchMerging.
    # Inputs are regenerated in the tests from (seed, shape, scale): x = randn(shape, gen(seed))*scale, dy = randn(y.shape, same gen)
    def block_case(tag, module, seed, shape, scale, wseed, clone_input=False):
        module.load_state_dict(seeded_state_dict(module.state_dict(), wseed))
        module.train()
        gb = torch.Generator().manual_seed(seed)
        xb = (torch.randn(*shape, generator=gb) * scale).requires_grad_(True)
        yb = module(xb.clone() if clone_input else xb)
        dyb = torch.randn(yb.shape, generator=gb)
        yb.backward(dyb)
        store[tag + "/meta"] = np.array([seed, wseed] + list(shape), dtype=np.int64)
        store[tag + "/scale"] = np.float64(scale)
        put(store, tag + "/y", yb.detach().numpy())
        put(store, tag + "/dx", xb.grad.numpy())
        for k, p in module.named_parameters():
            put(store, tag + "/grad/" + k, p.grad.numpy())

    block_case("tvblock", ref_tv.TinyViTBlock(192, (28, 28), 6, window_size=7, mlp_ratio=4.0, drop=0.0, drop_path=0.0), 31, (2, 784, 192), 0.7, 21)
    block_case("attn", ref_tv.Attention(384, 32, 12, attn_ratio=1, resolution=(14, 14)), 32, (2, 196, 384), 0.7, 22)
    block_case("mbconv", ref_tv.MBConv(96, 96, 4.0, torch.nn.GELU, 0.0), 33, (2, 96, 14, 14), 1.0, 23, clone_input=True)
    block_case("patchmerge", ref_tv.PatchMerging((28, 28), 192, 384, torch.nn.GELU), 34, (2, 784, 192), 1.0, 24)
    np.savez_compressed(os.path.join(OUT, "backbones.npz"), **store)
    print("backbones.npz:", len(store), "arrays,", os.path.getsize(os.path.join(OUT, "backbones.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
