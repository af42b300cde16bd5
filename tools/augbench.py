"""Throughput of the device-side augmentation (both views, rotation on) against the same chain in Pillow on the host."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from semiuhpe_amd.src import device_augment as D

n = 288                                            # images one C2 step consumes per GPU (32 labeled + 128 unlabeled x weak/strong views)
images = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device="cuda")
np.random.seed(0); torch.manual_seed(0)
t0 = time.perf_counter(); params = [D.draw_params(True, True) for _ in range(n)]; t_draw = time.perf_counter() - t0
t0 = time.perf_counter(); packed = D.pack_params(params, images.device, True); torch.cuda.synchronize(); t_pack = time.perf_counter() - t0
D.augment_views_packed(images, packed); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    D.augment_views_packed(images, packed)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / 20 * 1e-3
nbytes = n * 224 * 224 * 3 * (1 + 4) * 2          # uint8 in + fp32 out, two views (intermediates stay in L2 / MALL)
print(f"device: {n} images x 2 views in {dt*1e6:.0f} us = {n/dt:,.0f} img/s ({nbytes/dt/1e9:.0f} GB/s algorithmic); host: draw of the decisions {t_draw*1e3:.1f} ms, packing + upload {t_pack*1e3:.1f} ms")
try:
    from PIL import Image, ImageFilter
    host = images[:32].cpu().numpy()
    t0 = time.perf_counter()
    for k in range(32):
        p = params[k]; im = Image.fromarray(host[k])
        if p["flip"]: im = im.transpose(Image.FLIP_LEFT_RIGHT)
        if p["blur"]: im = im.filter(ImageFilter.BLUR)
        i, j, h, w = p["weak"]; a = np.asarray(im.crop((j, i, j + w, i + h)).resize((224, 224), Image.BILINEAR), np.float32) / 255
        r = im.rotate(p["angle"], center=(112, 112), expand=True); x0, y0 = r.size[0] // 2 - 112, r.size[1] // 2 - 112
        r = r.crop((x0, y0, x0 + 224, y0 + 224)); i, j, h, w = p["strong"]
        b = np.asarray(r.crop((j, i, j + w, i + h)).resize((224, 224), Image.BILINEAR), np.float32) / 255
    dtc = (time.perf_counter() - t0) / 32
    print(f"Pillow on one host core: {1/dtc:,.0f} img/s (x2 views, without the normalisation)")
except ImportError:
    pass
