"""Which envs MAY miss the tight HIP-vs-oracle bound of a heightfield step, decided from the ORACLE's own state (verdict of round 5:
an excuse must be a predicate, not a count).  The model has two discontinuities a last-bit difference between the two arithmetics can
land on different sides of:
  * a wheel making or breaking contact: the normal force max(k pen - c v_n, 0) has a kink at 0 and the implicit integrator's contact
    count jumps there -- the env is excusable if some wheel comes within FZ_EPS newtons of the switch at some sub-step;
  * a wheel's sample point crossing a cell line of the grid: the bilinear surface's normal jumps there -- excusable if some wheel comes
    within CELL_EPS cells of a line at some sub-step.
Both margins are recorded by the oracle (oracle/vehicle.py::substep `probe`, oracle/elev_step.py::ground_fn `probe`).  The thresholds are
~100 x what rounding can move a wheel (1e-6 m, 1e-4 N) and far below what the dynamics do in a sub-step."""
import numpy as np

FZ_EPS = 0.02        # N     (static load per wheel: 8.3 N)
CELL_EPS = 2e-3      # cells (0.1 mm at the 5 cm grid)


def contact_changed(probe, n):
    """bool [n]: the in-contact pattern of the four wheels changed between sub-steps of the step (a touch-down or a lift-off)"""
    masks = np.stack(probe["contact"])[:, :n]
    return (masks != masks[0]).any(0)


def explainable(probe, n):
    """bool [n]: the oracle's step of the env passed within the thresholds of one of the two discontinuities (a wheel that did make or
    break contact inside the step passed through the first)"""
    fz = np.asarray(probe.get("fz_margin", np.full(n, np.inf)))[:n]
    cell = np.asarray(probe.get("cell_margin", np.full(n, np.inf)))[:n]
    return (fz < FZ_EPS) | (cell < CELL_EPS) | contact_changed(probe, n)


WHEEL_ROWS, WHEEL_RADIUS = slice(13, 17), 0.05


def state_error(got, want, n, rows=21, tight=5e-4):
    """|got - want| in units of the bound: `tight` absolute + relative on every row of the state.  The four wheel-spin rows are held to
    2 x tight of CONTACT SPEED instead: a spin is a contact speed over r = 0.05 m (1e-3 m/s is 2e-2 rad/s), and the spin solve divides by
    A0 + K r^2 with K the tyre's secant stiffness -- the one place where the step's rounding is amplified (measured, device physics
    compiled for the host vs this oracle, 4096 random states, 10 x 20 ms: spins to 1.8e-2 rad/s where every body row holds 1e-4)"""
    atol = np.full((rows, 1), tight)
    rtol = np.full((rows, 1), tight)
    atol[WHEEL_ROWS], rtol[WHEEL_ROWS] = 2 * tight / WHEEL_RADIUS, 2 * tight
    return np.abs(got[:rows, :n] - want[:rows, :n]) / (atol + rtol * np.abs(want[:rows, :n]))


def check_state(got, want, probe, n, ok, rows=21, tight=5e-4, loose=400.0, where=""):
    """every env in `ok` holds rows [0, rows) of the state to `tight` (state_error) -- except envs the predicate explains, which are held
    to `loose` x that bound and must be few.  -> (mask of the envs that met the tight bound, number excused)"""
    err = state_error(got, want, n, rows, tight)
    touchy = (err.max(0) > 1.0) & ok
    ex = explainable(probe, n)
    unexplained = touchy & ~ex
    if unexplained.any():
        e = int(np.argmax(np.where(unexplained, err.max(0), 0)))
        raise AssertionError(f"{where}: {int(unexplained.sum())} env(s) miss the bound with no discontinuity in reach; worst env {e}: "
                             f"{float(err[:, e].max()):.2f} x the bound in row {int(err[:, e].argmax())}, fz_margin "
                             f"{float(np.asarray(probe['fz_margin'])[e]):.4f} N, cell_margin {float(np.asarray(probe.get('cell_margin', [np.inf] * n))[e]):.5f}")
    assert err[:, touchy].max(initial=0) < loose, (where, float(err[:, touchy].max()))
    return ok & ~touchy, int(touchy.sum())
