import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "store_patterns.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "store_patterns.hip")])
lib = ctypes.CDLL(so)
lib.store_pattern.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for rows, cols in ((25088, 960), (25088, 160), (6272, 1536), (401408, 192), (100352, 256)):
    y = torch.empty(rows, cols, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for which, name in ((0, "mfma dword 2x128B"), (1, "wave f4 8x128B"), (2, "block f4 2x512B")):
        for _ in range(3):
            lib.store_pattern(which, y.data_ptr(), rows, cols, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.store_pattern(which, y.data_ptr(), rows, cols, st)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 20 * 1e-3
        print(f"{rows}x{cols} {name:20s} {dt*1e6:8.1f} us  {rows*cols*4/dt/1e9:7.0f} GB/s", flush=True)
