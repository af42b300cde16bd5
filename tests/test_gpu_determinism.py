"""Run-to-run determinism of the training step: every reduction in the library has a fixed order (no atomics), so two processes that run the same seeded
SSL steps with the same tile choices must end on bit-identical student and teacher weights.  A difference means a race -- a hand-counted s_waitcnt of the
Winograd kernels' chunk loop off by one, an LDS image read before its barrier, two streams touching one workspace -- which no tolerance test would notice.
(With the autotuner MEASURING tiles at run time two runs may pick different tiles for a shape it has no database entry for, and then differ legitimately:
the test pins the choice with SUHPE_AUTOTUNE=0 and lets the heuristics prefer the Winograd tiles, SUHPE_WINOGRAD=2.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("net,steps,size", [("effinetv2", 6, 128), ("repvgg", 4, 128), ("tinyvit", 3, 224)])
def test_two_runs_end_on_identical_weights(net, steps, size):
    env = dict(os.environ, SUHPE_AUTOTUNE="0", SUHPE_WINOGRAD="2", DET_B="8", DET_SIZE=str(size))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "determinism_soak.py"), net, str(steps)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.strip().endswith("IDENTICAL"), r.stdout[-1500:]
    assert r.stdout.count("DIGEST") == 2
