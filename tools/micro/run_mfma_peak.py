import ctypes, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "mfma_peak.hip")])
lib = ctypes.CDLL(so)
lib.mfma_peak.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.empty(256 * 8 * 256 * 4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 2000
for chains in (1, 2, 4):
    for per_cu in (1, 2, 4, 8):          # 256-thread blocks: one wave per SIMD each
        blocks = 256 * per_cu
        lib.mfma_peak(chains, blocks, iters, out.data_ptr(), st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mfma_peak(chains, blocks, iters, out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 5 * 1e-3
        flops = blocks * 4 * iters * 16 * 4096.0
        print(f"chains={chains} waves/SIMD={per_cu}: {flops/dt/1e12:7.1f} TF/s ({dt*1e3:.2f} ms)", flush=True)
