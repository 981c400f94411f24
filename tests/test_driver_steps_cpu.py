"""CPU: the oracle's restatement of the swint_opt routines against the reference itself
(tests/golden/swinterp_day.npz = outputs of update_swinterp_parameters / interp_sw_radiation compiled from
module_radiation_driver.F:2668-2795, see tools/make_golden_swinterp.py), and the bucket accumulation."""
import os

import numpy as np

from oracle import oracle_driver as od

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "swinterp_day.npz")
# pow / log come from different math libraries (flang runtime vs glibc): agreement to a few ulp, not bitwise
RTOL, ATOL = 5e-13, 1e-12


def run_day(update, apply, g, real4=False):
    """Feed the golden day through `update` / `apply`; yields (step, state, interpolated fields)."""
    n, nstep = int(g["n"]), int(g["nstep"])
    state = {k: np.zeros(n) for k in od.UPDATE_IO}
    for s in range(nstep):
        inp = {k: np.ascontiguousarray(g["in_" + k][s]) for k in od.UPDATE_IN}
        update(inp, state)
        out = {"swdown": inp["swdown"].copy(), "swddir": inp["swddir"].copy(), "swddni": np.zeros(n),
               "swddif": np.zeros(n), "gsw": np.zeros(n)}
        apply(state, np.ascontiguousarray(g["in_coszen_loc2"][s]), np.ascontiguousarray(g["in_albedo"][s]), out)
        yield s, {k: v.copy() for k, v in state.items()}, out


def check_day(update, apply, g, rtol, atol):
    for s, state, out in run_day(update, apply, g):
        for k in od.UPDATE_IO:
            np.testing.assert_allclose(state[k], g["ref_" + k][s], rtol=rtol, atol=atol, err_msg=f"step {s} {k}")
        for k in od.APPLY_OUT:
            np.testing.assert_allclose(out[k], g["ref_o_" + k][s], rtol=rtol, atol=atol, err_msg=f"step {s} {k}")


def test_swinterp_oracle_matches_reference_day():
    g = np.load(GOLD)
    # the fixture exercises every branch: clamped and free exponents, unchanged zenith angle, night
    bb = g["ref_bb"]
    assert ((bb == -0.5) | (bb == 2.5)).sum() > 50 and (bb == 0).sum() > 50
    assert ((bb != -0.5) & (bb != 2.5) & (bb != 0)).sum() > 100
    assert (g["ref_o_swdown"] == 0).any() and (g["ref_o_swdown"] > 0).any()
    check_day(lambda i, st: od.swinterp_update(i, st), lambda st, c, a, o: od.swinterp_apply(st, c, a, o), g, RTOL, ATOL)


def test_accumulate_oracle():
    rng = np.random.default_rng(5)
    f, a = rng.uniform(0, 400, 100), rng.uniform(0, 1e6, 100)
    ref = a + f * 72.0
    od.accumulate(f, a, 72.0)
    np.testing.assert_array_equal(a, ref)
