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


def test_winograd_tile_selection_is_host_logic():
    """the Winograd tiles are menu entries: which geometries may pin them, what the transformed filter and the filter-gradient workspace
    take, and that the pin is what suhpe_conv2d_wino_pinned reports -- all decided on the host, no GPU needed"""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    lib.suhpe_conv2d_wino_filter_bytes.restype = ctypes.c_int64
    lib.suhpe_conv2d_wgrad_workspace_bytes.restype = ctypes.c_int64
    names = {p: [lib.suhpe_conv2d_tile_name(p, i).decode() for i in range(lib.suhpe_conv2d_tile_count(p))] for p in (0, 1, 2)}
    assert names[0] == names[1]
    nt, tn = names[0].index("wino_nt_kernel<64,64>"), names[2].index("wino_wgrad_kernel<64,64>")
    ok = _lib.ConvDesc(32, 28, 28, 128, 28, 28, 128, 3, 3, 1, 1, 1)            # 3x3, stride 1, pad 1
    for bad in (_lib.ConvDesc(32, 28, 28, 128, 14, 14, 128, 3, 3, 2, 1, 1),    # stride 2
                _lib.ConvDesc(32, 28, 28, 128, 28, 28, 128, 1, 1, 1, 0, 1),    # pointwise
                _lib.ConvDesc(32, 28, 28, 3, 28, 28, 24, 3, 3, 1, 1, 1),       # 3 input channels: not a multiple of 4
                _lib.ConvDesc(32, 7, 7, 512, 7, 7, 512, 3, 3, 1, 1, 1)):       # odd image size (2 x 2 output tiles need even heights and widths)
        for p, t in ((0, nt), (2, tn)):
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(bad), p, t) == -1, (p, bad.cin, bad.stride)
            assert lib.suhpe_conv2d_wino_pinned(ctypes.byref(bad), p) == 0
    try:
        for p in (0, 1):
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(ok), p, nt) == 0
            assert lib.suhpe_conv2d_wino_pinned(ctypes.byref(ok), p) == 1
            # 16 positions x (channels rounded to super-chunks of 16) x 64-wide cout blocks, fp32
            assert lib.suhpe_conv2d_wino_filter_bytes(ctypes.byref(ok), p) == 16 * 128 * 128 * 4
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(ok), 2, tn) == 0
        splits = lib.suhpe_conv2d_wgrad_splits(ctypes.byref(ok))
        blocks = (128 // 64) * (128 // 64)
        assert 1 <= splits <= 256 // blocks                                    # one generation of workgroups on 256 CUs
        assert lib.suhpe_conv2d_wgrad_workspace_bytes(ctypes.byref(ok)) >= splits * 16 * 128 * 128 * 4
    finally:
        for p in (0, 1, 2):
            assert lib.suhpe_conv2d_set_tile(ctypes.byref(ok), p, -1) == 0
    assert lib.suhpe_conv2d_wino_pinned(ctypes.byref(ok), 0) in (0, 1)          # back to the heuristic (SUHPE_WINOGRAD decides)


def test_consuming_finalizer_rejects_inconsistent_segments():
    """suhpe_bn_stats_from_partials_consume: one segment takes every record and no boundary record; two segments need records or a boundary record
    for the second one (host-side validation, nothing is launched)"""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    f = lib.suhpe_bn_stats_from_partials_consume
    buf = ctypes.c_void_p(16)                                 # never dereferenced: every call below fails validation
    args = dict(eps=ctypes.c_float(1e-3), mom=ctypes.c_float(0.1))

    def call(tiles, first, boundary, rows, rows_a, c=8):
        return f(buf, tiles, first, boundary, ctypes.c_int64(rows), ctypes.c_int64(rows_a), c, args["eps"], args["mom"], buf, buf, None, None, None, None)
    assert call(10, 10, None, 640, 320) == -1                 # two segments, no record and no boundary record for the second
    assert b"consume" in lib.suhpe_last_error()
    assert call(10, 4, None, 640, 640) == -1                  # one segment must take every record
    assert call(10, 10, buf, 640, 640) == -1                  # ... and has no boundary record
    assert call(10, 11, None, 640, 320) == -1                 # more first-segment records than records
    assert call(0, 0, None, 640, 640) == -1
    assert call(10, 10, None, 640, 0) == -1


def test_collective_library_override_that_does_not_load_is_an_error():
    """SUHPE_RCCL_LIB names another build of the collective library (the host-memory transport of the two-rank tests).  A path that does not
    load must leave the communicator UNAVAILABLE with a message that names the path -- never a silent fall back to the system RCCL
     -- and must not disturb the rest of the library.  Own process: the library resolves RCCL once."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes\n"
            "from semiuhpe_amd import _lib\n"
            "lib = _lib.load()\n"
            "assert lib.suhpe_comm_available() == 0\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "assert lib.suhpe_comm_unique_id(buf) == -1\n"
            "msg = lib.suhpe_last_error()\n"
            "assert b'/nonexistent/libnot_rccl.so' in msg and b'did not load' in msg, msg\n"
            "assert lib.suhpe_abi_version() == 1\n"
            "print('ok')\n")
    env = dict(os.environ, SUHPE_RCCL_LIB="/nonexistent/libnot_rccl.so", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_last_kernel_name_is_exported_and_empty_before_any_launch():
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from semiuhpe_amd import _lib\nassert _lib.load().suhpe_last_kernel_name() == b''\nprint('ok')\n"
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_pinned_tile_without_an_instantiation_is_an_error_on_every_entry_point():
    """the narrow LDS-DMA tile writes 32-row statistics records that no other tile writes, so a call it cannot take (a bias)
    with statistics asked for must come back as SUHPE_EINVAL -- from suhpe_conv2d_fwd_wino too, which used to return SUHPE_OK with y and the records
    unwritten.  Decided on the host before anything is launched: no GPU needed (the pointers are never dereferenced)."""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
    d = _lib.ConvDesc(4, 56, 56, 192, 56, 56, 48, 1, 1, 1, 0, 1)               # FusedMBConv's project conv: 192 -> 48, pointwise
    P = ctypes.c_void_p(1 << 20)                                               # 16-byte aligned, never read
    assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names.index("gemm_nt_dma<256,64>")) == 0
    try:
        assert lib.suhpe_conv2d_fwd_stats_tile_rows(ctypes.byref(d)) == 32
        rc = lib.suhpe_conv2d_fwd_wino(ctypes.byref(d), P, P, None, P, P, P, None, ctypes.c_int64(0), None)      # bias AND statistics
        assert rc == -1, rc
        msg = lib.suhpe_last_error()
        assert b"suhpe_conv2d_fwd_wino" in msg and b"no instantiation" in msg, msg
    finally:
        assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, -1) == 0
