"""python tools/micro/run_coissue_probe.py  (on the GPU box): time of a 16-MFMA loop with NV vector-ALU instructions behind each MFMA, one wavefront per
SIMD; then two wavefronts per SIMD with split roles (MFMA-only / VALU-only workgroups)."""
import ctypes, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "coissue_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "coissue_probe.hip")])
lib = ctypes.CDLL(so)
lib.coissue.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.empty(256 * 8 * 256 * 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 4000
def run(nv, blocks, split):
    lib.coissue(nv, blocks, iters, split, out.data_ptr(), st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        lib.coissue(nv, blocks, iters, split, out.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
for per_cu in (1, 2):
    for nv in (0, 2, 4, 6, 8, 12):
        t = run(nv, 256 * per_cu, 0)
        cyc = t * 1e-3 * 2.4e9 / (iters * 16)
        print(f"same wave, {per_cu} wave/SIMD, NV={nv:2d} per MFMA: {t:7.3f} ms  ~{cyc:5.1f} cycles per (MFMA + NV VALU) at 2.4 GHz", flush=True)
for nv in (4, 8, 12):
    t = run(nv, 512, 1)
    cyc = t * 1e-3 * 2.4e9 / (iters * 16)
    print(f"split roles (1 MFMA wave + 1 VALU wave per SIMD), NV={nv:2d}: {t:7.3f} ms  ~{cyc:5.1f} cycles per 16x16x4 slot", flush=True)
