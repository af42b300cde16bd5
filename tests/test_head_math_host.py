"""The scalar head math that runs inside the gfx950 kernels (semiuhpe_amd/csrc/head_math.h) compiled for the
host with g++ and checked against the reference's golden vectors.  CPU-only; the GPU parity tests call the
same math through the C-ABI (tests/test_gpu_head.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_build", "head_host.cpp")
SO = os.path.join(HERE, "host_build", "libhead_host.so")
G = np.load(os.path.join(HERE, "golden", "head.npz"))
NAMES = [str(n) for n in G["names"]]


@pytest.fixture(scope="module")
def lib():
    hdr = os.path.join(HERE, "..", "semiuhpe_amd", "csrc", "head_math.h")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", SO, SRC])
    return ctypes.CDLL(SO)


def fp(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)


def rel(x, ref):
    return np.abs(np.asarray(x, np.float64) - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", NAMES)
def test_head_math_vs_reference(lib, name):
    A, A1, R = G[name + "/A"], G[name + "/A1"], G[name + "/R"]
    n = A.shape[0]
    loss, dA, Rest = np.zeros(n, np.float32), np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32)
    lib.host_nll(fp(A), fp(R), n, ctypes.c_double(1.025), fp(loss), fp(dA), fp(Rest))
    assert rel(loss, G[name + "/nll"]) < 1e-5
    assert rel(dA, G[name + "/nll_dA"]) < 1e-5
    assert rel(Rest.reshape(n, 3, 3), G[name + "/Rest"]) < 5e-6
    H = np.zeros(n, np.float32)
    lib.host_entropy(fp(A), n, fp(H))
    assert rel(H, G[name + "/entropy"]) < 5e-5
    ce, dA2 = np.zeros(n, np.float32), np.zeros((n, 9), np.float32)
    lib.host_ce(fp(A1), fp(A), n, fp(ce), fp(dA2))
    assert rel(ce, G[name + "/ce"]) < 1e-5
    assert rel(dA2, G[name + "/ce_dA2"]) < 5e-5


def test_svd_properties(lib):
    rng = np.random.default_rng(0)
    A = (rng.standard_normal((256, 9)) * rng.choice([1e-3, 1.0, 50.0], (256, 1))).astype(np.float32)
    A[:4] = 0
    A[4:8, 3:] = 0                       # rank one
    n = A.shape[0]
    U, S, V, R = (np.zeros((n, 9), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 9), np.float32),
                  np.zeros((n, 9), np.float32))
    lib.host_svd(fp(A), n, fp(U), fp(S), fp(V), fp(R))
    U3, V3, R3 = U.reshape(n, 3, 3).astype(np.float64), V.reshape(n, 3, 3).astype(np.float64), R.reshape(n, 3, 3)
    rec = U3 @ (S[:, :, None] * V3.transpose(0, 2, 1))
    assert np.abs(rec - A.reshape(n, 3, 3)).max() < 1e-5 * max(1.0, np.abs(A).max())
    assert np.allclose(np.linalg.det(U3), 1, atol=1e-5) and np.allclose(np.linalg.det(V3), 1, atol=1e-5)
    assert np.allclose(np.linalg.det(R3.astype(np.float64)), 1, atol=1e-5)
    assert (S[:, 0] >= S[:, 1] - 1e-6).all() and (S[:, 1] >= np.abs(S[:, 2]) - 1e-6).all()
    sv = np.linalg.svd(A.reshape(n, 3, 3).astype(np.float64), compute_uv=False)
    assert np.abs(np.abs(S) - sv).max() < 1e-5 * max(1.0, sv.max())


def test_euler_geodesic(lib):
    D = np.load(os.path.join(HERE, "golden", "euler.npz"))
    R = D["R"]
    n = R.shape[0]
    for fr in (0, 1):
        e = np.zeros((n, 3), np.float32)
        lib.host_euler(fp(R), n, fr, fp(e))
        assert np.abs(e - D[f"euler_full{fr}"]).max() < 2e-6
    out = np.zeros(n, np.float32)
    lib.host_geodesic_deg(fp(R), fp(D["R2"]), n, fp(out))
    assert np.abs(out - D["geodesic_deg"]).max() < 1e-3


def test_head_math_under_sanitizers(tmp_path):
    """AddressSanitizer + UndefinedBehaviorSanitizer over the head math the gfx950 kernels share with this host build
    (random, degenerate and extreme-scale inputs through every entry point; GPU sanitizers are not available on the pool)"""
    exe = str(tmp_path / "head_sanitize")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(HERE, "host_build", "head_sanitize_main.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HEAD_SANITIZE_OK non-finite 0" in r.stdout, r.stdout + r.stderr[-2000:]
