"""The tile autotuner never enters a collective: with more than one rank a shape outside the tile database keeps the heuristic tile
(the broadcast of rank 0's winner hung jobs whose ranks see different shapes, e.g. a ragged validation deal)."""
import ctypes

import torch
import torch.distributed as dist

from semiuhpe_amd import kernels as K


def test_autotune_measures_nothing_and_calls_no_collective_with_two_ranks(monkeypatch):
    calls = []
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 2)
    for name in ("broadcast", "all_reduce", "all_gather", "barrier", "broadcast_object_list"):
        monkeypatch.setattr(dist, name, lambda *a, _n=name, **k: calls.append(_n))
    monkeypatch.setattr(K, "_tuned", set())
    monkeypatch.setattr(K, "_TILE_DB", {})                       # nothing known: the single-rank path would time the whole menu
    d = K.conv_desc((5, 6, 6, 8), (12, 8, 3, 3), 1, 1, 1)
    ran = []
    K._autotune(d, 0, lambda: ran.append(1))
    assert not ran, "no trial launch with more than one rank"
    assert not calls, f"collectives inside the autotuner: {calls}"
    assert K._TILE_DB == {}                                      # and nothing is recorded as a measured winner
