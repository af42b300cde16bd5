"""semiuhpe_amd — MI355X (gfx950) native hot path of SemiUHPE's teacher-student training step.

Layout
    csrc/        hand-written HIP kernels + the C ABI declared in include/suhpe.h  (libsuhpe.so)
    _lib.py      ctypes binding of that C ABI (fails loudly if the library is missing)
    kernels.py   tensor-level wrappers (raw device pointers on the current HIP stream)
    layers.py    layers with explicit forward/backward over those kernels (no autograd inside the backbone)
    src/         host-side mirror of the reference's Python interface (src/agent.py, src/networks.py, ...)

Importing this package does not need a GPU; running any op does.
"""
__version__ = "0.1.0"
