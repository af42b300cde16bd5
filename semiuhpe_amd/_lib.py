"""ctypes binding of libsuhpe.so — the C ABI declared in include/suhpe.h.

The prototypes are parsed from the header itself, so the header is the single source of truth: every function it
declares must be exported by the library (checked at load time and by tests/test_cabi.py).

There is NO fallback: if the library cannot be loaded, or an op is invoked without a GPU, a RuntimeError is raised.
Error codes map to the exception types of the reference's Python API: SUHPE_EINVAL -> ValueError,
everything else -> RuntimeError.
"""
import ctypes
import os
import re

import torch  # imported first on purpose: libsuhpe.so must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUHPE_LIB") or os.path.join(_HERE, "csrc", "libsuhpe.so")      # SUHPE_LIB: a variant build (semiuhpe_amd.build.build_variant)
HEADER_PATH = os.path.join(_HERE, "..", "include", "suhpe.h")


class ConvDesc(ctypes.Structure):
    """suhpe_conv_t"""
    _fields_ = [(k, ctypes.c_int) for k in ("n", "h", "w", "cin", "ho", "wo", "cout", "r", "s", "stride", "pad", "groups")]


_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
    "suhpe_stream_t": ctypes.c_void_p, "suhpe_comm_t": ctypes.c_void_p, "void": None,
}


def _ctype_of(decl):
    decl = decl.replace("const ", "").strip()
    if "*" in decl:
        base = decl.split("*")[0].strip()
        if base == "suhpe_conv_t":
            return ctypes.POINTER(ConvDesc)
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    return _CTYPES[decl.split()[0]]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in suhpe.h"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(suhpe_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or "}" in ret:
            continue
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a   # drop the parameter name
                argtypes.append(_ctype_of(a))
        protos[name] = (_ctype_of(ret), argtypes)
    return protos


_lib = None
_protos = None


def load():
    """Load libsuhpe.so (once).  Raises RuntimeError when it is missing — there is no CPU path."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"semiuhpe_amd: {LIB_PATH} not found. Build it with `python -m semiuhpe_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    missing = [n for n in protos if not hasattr(lib, n)]
    if missing:
        raise RuntimeError(f"semiuhpe_amd: libsuhpe.so does not export {missing} declared in include/suhpe.h")
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib, _protos = lib, protos
    return lib


def last_error():
    return load().suhpe_last_error().decode()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("semiuhpe_amd: no HIP device visible; the hot path has no CPU fallback "
                           "(the CPU oracle under oracle/ is test infrastructure only)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw-handle query: building a torch.cuda.Stream
    object per launch costs more host time than the launch itself)."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on the current stream (appended automatically when the prototype ends
    with suhpe_stream_t) and raise on a non-zero code."""
    lib = load()
    fn = getattr(lib, name)
    if fn.argtypes and fn.argtypes[-1] is ctypes.c_void_p and len(args) == len(fn.argtypes) - 1:
        args = args + (stream_ptr(),)
    rc = fn(*args)
    if rc != 0:
        msg = f"{name} failed ({rc}): {lib.suhpe_last_error().decode()}"
        raise ValueError(msg) if rc == -1 else RuntimeError(msg)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------ workspaces
_ws = {}


def workspace(tag, nbytes, device=None):
    """Grow-only scratch buffer per (tag, device, stream).  Launches on one stream are ordered, so a tag can be reused by
    consecutive kernels; a second stream (the teacher pass runs beside the student's) gets its own buffers."""
    device = device if device is not None else torch.cuda.current_device()
    key = (tag, torch.device("cuda", device) if isinstance(device, int) else device, stream_ptr())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=key[1])
        _ws[key] = buf
    return buf
