"""The torch-CPU port timed as bench.py's cpu_baseline must compute the same numbers as the reference (golden)."""
import numpy as np
import torch

from oracle import torch_mdp as T


def test_torch_port_matches_reference_golden(golden):
    for tag in ("n256", "edges"):
        g = {k: torch.from_numpy(v) for k, v in golden(f"drift_mdp_{tag}").items()}
        tol = dict(rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(T.side_slip(g["lin_vel_b"]), g["side_slip"], **tol)
        np.testing.assert_allclose(T.vel_dist(g["lin_vel_b"]), g["vel_dist"], **tol)
        np.testing.assert_allclose(T.turn_left_go_right(g["joint_pos"][:, 0:2], g["ang_vel_b"]), g["turn_left_go_right"], **tol)
        np.testing.assert_allclose(T.energy_through_turn(g["pos"], g["lin_vel_b"]), g["energy_through_turn"], **tol)
        np.testing.assert_allclose(T.cross_track_dist(g["pos"]), g["cross_track_dist"], **tol)
        assert torch.equal(T.cart_off_track(g["pos"]), g["cart_off_track"])


def test_torch_port_mdp_step_shapes(golden):
    g = {k: torch.from_numpy(v) for k, v in golden("drift_mdp_n256").items()}
    n = g["pos"].shape[0]
    obs, rew, term, trunc, st, wt = T.mdp_step(g["pos"], g["quat"], g["lin_vel_b"], g["ang_vel_b"], g["ang_vel_w"],
                                              g["joint_pos"][:, 0:2], g["actions"], torch.zeros(n, dtype=torch.int32),
                                              [10., -5., 40., 0., 20., -50., -5000.], torch.tensor([3.0, 0.488]),
                                              torch.zeros(2))
    assert obs.shape == (n, 14) and rew.shape == (n,) and term.dtype == torch.bool and st.shape == (n, 2)
