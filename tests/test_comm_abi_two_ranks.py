"""Two ranks through the suhpe_comm_* C ABI without GPUs: libsuhpe.so resolves the collective library with dlopen, so
SUHPE_RCCL_LIB points it at a host-memory transport (tests/host_build/hostmem_transport.c -- test infrastructure, NOT RCCL) and two processes
drive unique_id / init / allreduce (sum, average) / allgather / broadcast (root != 0) / zero-length calls / error mapping / destroy; a second
pair runs semiuhpe_amd.src.distributed in SUHPE_COMM=native mode (gloo as the host channel, every collective of the step through the ABI)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_build", "hostmem_transport.c")
LIB = os.path.join(HERE, "host_build", "libhostmem_transport.so")


def _transport():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", SRC, "-o", LIB, "-lrt", "-pthread"])
    return LIB


def _run(mode, world, arg, extra_env=None):
    env = dict(os.environ, SUHPE_RCCL_LIB=_transport(), PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_abi_worker.py"), mode, str(r), str(world), arg], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                   # exact child PIDs only
            raise
        outs.append((p.returncode, out))
    for rc, out in outs:
        assert rc == 0, out[-3000:]
    return outs


@pytest.mark.parametrize("world", [2, 3])
def test_comm_abi_with_several_ranks_over_the_host_transport(tmp_path, world):
    outs = _run("abi", world, str(tmp_path / "unique_id.bin"))
    assert all("abi ok" in o for _, o in outs)


def test_distributed_native_mode_is_the_only_communicator_of_the_step(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = _run("dist", 2, str(port), {"SUHPE_COMM": "native", "SUHPE_COMM_HOST_BUFFERS": "1"})
    assert all("distributed native ok" in o for _, o in outs)
