"""SyncBN (optional mode): two ranks with layers.SYNC_BN, each on half of a batch, must reproduce the
single-process forward / backward on the whole batch -- outputs, input gradients, summed parameter gradients, running statistics."""
import os
import subprocess
import sys

import pytest
import torch

from parity_utils import rel_err

pytestmark = pytest.mark.gpu


def test_two_ranks_with_syncbn_equal_one_process_on_the_whole_batch(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "tests", "syncbn_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    sys.path.insert(0, os.path.join(root, "tests"))
    import syncbn_worker as W
    blk = W.build()
    x, dy = W.data()
    y, dx = W.run(x, dy, blk)
    assert rel_err(torch.cat([r0["y"], r1["y"]]), y) < 1e-5
    assert rel_err(torch.cat([r0["dx"], r1["dx"]]), dx) < 2e-4
    for n, p in blk.named_parameters():
        assert rel_err(r0["grads"][n], p.grad) < 2e-4, n
        assert torch.equal(r0["grads"][n], r1["grads"][n])
    for n, b in blk.named_buffers():
        if b.dtype == torch.float32:
            assert rel_err(r0["buffers"][n], b) < 1e-5 and rel_err(r1["buffers"][n], b) < 1e-5, n
