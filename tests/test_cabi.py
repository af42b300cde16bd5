"""CPU-side checks of the drop-in boundary: libsuhpe.so loads through ctypes, exports every symbol that
include/suhpe.h declares, and rejects bad arguments with the documented error codes (no GPU work is launched)."""
import ctypes

import pytest


def test_library_exports_every_declared_symbol():
    from semiuhpe_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    lib = _lib.load()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.suhpe_abi_version() == 1


def test_argument_validation_without_gpu():
    from semiuhpe_amd import _lib
    lib = _lib.load()
    assert lib.suhpe_fisher_entropy(None, 4, None, None) == -1
    assert b"suhpe_fisher_entropy" in lib.suhpe_last_error()
    d = _lib.ConvDesc(1, 8, 8, 16, 9, 9, 16, 3, 3, 1, 1, 1)      # wrong ho/wo
    assert lib.suhpe_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, None) == -1
    assert b"ho/wo" in lib.suhpe_last_error()
    assert lib.suhpe_rot_adjust(16, 16, 1, 7, 16, None) == -1
    with pytest.raises(ValueError):
        _lib.call("suhpe_rot_adjust", 16, 16, 1, 7, 16, None)


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from semiuhpe_amd import kernels
    with pytest.raises(RuntimeError, match="no CPU"):
        kernels.fisher_entropy(torch.zeros(4, 9))
