"""Pin the elevation / visual oracle terms to the golden vectors produced by the reference's own functions."""
import numpy as np

from oracle import elev_mdp as E


def test_elevation_terms_match_reference(golden):
    g = golden("elevation_mdp")
    tol = dict(rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(E.world_height_map(g["sensor_pos_w"][:, 2], g["ray_hits_z"], g["pos"][:, 2]),
                               g["world_height_map"], rtol=1e-6, atol=4e-6)   # (z+20) - hit: fp32 cancellation at ~20
    np.testing.assert_allclose(E.goal_relative_xyz(g["pos"], g["command"]), g["goal_relative_xyz"], **tol)
    np.testing.assert_allclose(E.goal_progress_rate(g["pos"], g["lin_vel_w"], g["command"]), g["goal_progress_rate"],
                               rtol=1e-5, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(E.higher_elevation(g["pos"], g["lin_vel_b"]), g["higher_elevation"], **tol)
    np.testing.assert_array_equal(E.is_falling_penalty(g["lin_vel_b"]), g["is_falling_penalty"])
    np.testing.assert_allclose(E.forward_vel(g["lin_vel_b"]), g["forward_vel"], **tol)
    np.testing.assert_array_equal(E.stuck(g["lin_vel_b"], g["joint_vel"][:, 2:6]), g["stuck"])
    np.testing.assert_allclose(E.upright_penalty(g["quat"], 60.0), g["upright_penalty"], rtol=1e-4, atol=2e-3)  # acos, degrees
    ok = np.abs(g["upright_penalty"]) > 1e-2                                   # away from the 60 deg tie
    np.testing.assert_array_equal(E.upright_bool(g["quat"])[ok | (g["upright_penalty"] == 0)],
                                  g["upright_bool"][ok | (g["upright_penalty"] == 0)])
    np.testing.assert_array_equal(E.close_to_goal(g["pos"], g["command"]), g["close_to_goal"])
    np.testing.assert_array_equal(g["weights"], np.array([200.0, 5000.0, 0.0, -200.0], np.float32))
    assert g["close_to_goal"].sum() >= 4 and g["stuck"].sum() >= 1 and g["upright_bool"].sum() >= 10   # branches are exercised
