"""Row d2 of the scope table (BASELINE.json's metric, second half): AFLW2000 Euler-MAE parity of the HIP path
with the CPU oracle on the 1 964-image synthetic evaluation set -- rotation matrices within 1e-4, pitch / yaw / roll / mean MAE
within 0.05 degrees -- after K = 3 identical SSL warm-up steps on both paths (tools/mae_parity.py holds the protocol)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

R_TOL, MAE_TOL = 1e-4, 0.05          # north_star: rotation matrices within 1e-4, Euler MAE within +-0.05 degrees


def test_euler_mae_parity_full_set():
    import mae_parity
    out = mae_parity.run(n_images=mae_parity.EVAL_IMAGES, warm_steps=3, train_batch=(8, 16), size=224, network="effinetv2")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "mae_parity.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    assert out["images"] == 1964
    same, own = out["same_weights"], out["own_trajectory"]
    # the eval path on identical weights: the north star's bars
    assert same["max_abs_dR"] <= R_TOL, same
    assert all(v <= MAE_TOL for v in same["d_mae_deg"].values()), same
    assert out["ema_same_weights_max_abs_dR"] <= R_TOL
    # each path on the weights it trained itself for K steps: the MAE bar holds; rotation matrices carry K steps of fp32 training
    # noise through Adam's sign-like first updates (the reference's own trajectories differ by as much between two runs)
    assert own["max_abs_dR"] <= 5e-3, own
    assert all(v <= MAE_TOL for v in own["d_mae_deg"].values()), own
    wt = out["within_tolerance"]
    assert wt["same_weights"] and wt["ema_same_weights"] and wt["own_trajectory_mae"], wt
    # (own_trajectory's rotation matrices are not a tolerance row any more: they are held to the CONTROL trajectory below)
    assert "own_trajectory_R" not in wt and "own_trajectory_vs_control" in out["note"]


def test_own_trajectory_drift_is_held_to_the_control():
    """drift as a TEST instead of a quoted file.  Steps 1...8 of the drift_trend protocol (tools/mae_parity.py) at a reduced
    batch: both paths train from the same conditioned initial state, each on its OWN weights, and so does a CONTROL -- the CPU oracle against a copy
    of itself whose initial weights differ by one unit in the last place.  After steps 1, 2, 3, 5, 8 the HIP path must be no further from the oracle
    than 2 x the control is (max |dR| over the evaluated images; the two curves coincide at the BASELINE batch: profiles/drift_trend.json) and
    the Euler MAE must agree within 0.05 degrees.  A kernel that adds error beyond fp32 summation-order noise shows up here as a curve ABOVE the
    control's, which no fixed bar on a chaotic quantity can tell from the chaos."""
    import mae_parity
    doc = mae_parity.drift_trend(steps=8, at=(1, 2, 3, 5, 8), train_batch=(8, 16), n_images=64, size=224, network="effinetv2", threads=32)
    with open(os.path.join(ROOT, "gpurun_out", "drift_test.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    assert [r["step"] for r in doc["rows"]] == [1, 2, 3, 5, 8]
    for r in doc["rows"]:
        c = r["control_oracle_vs_oracle_1ulp"]
        assert r["max_abs_dR"] <= 2.0 * c["max_abs_dR"], (r["step"], r["max_abs_dR"], c["max_abs_dR"])
        assert r["mean_abs_dR"] <= 2.0 * c["mean_abs_dR"], (r["step"], r["mean_abs_dR"], c["mean_abs_dR"])
        assert all(v <= MAE_TOL for v in r["d_mae_deg"].values()), r
    assert doc["rows"][-1]["control_oracle_vs_oracle_1ulp"]["max_abs_dR"] > 1e-4        # (the control itself leaves the identical-weights bar: chaos)
