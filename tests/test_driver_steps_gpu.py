"""GPU: the driver-side neighbours (bucket accumulation, swint_opt update / interpolation) through the C ABI
against the oracle and against the reference's own outputs (golden day), device and host pointers, REAL*8 and
REAL*4 storage, tile inside a larger memory window."""
import numpy as np
import pytest

from oracle import oracle_driver as od
from test_driver_steps_cpu import GOLD, check_day

pytestmark = pytest.mark.gpu


def _dims(ni, nj, halo=0):
    return dict(ims=1 - halo, ime=ni + halo, jms=1 - halo, jme=nj + halo, its=1, ite=ni, jts=1, jte=nj)


def _hip_update(wa, dims, dev):
    import torch

    def f(inp, state):
        arrs = {**inp, **state}
        if dev:
            t = {k: torch.from_numpy(v.reshape(1, -1).copy()).cuda() for k, v in arrs.items()}
            wa.driver_steps.update_swinterp_parameters(dims, **t)
            for k in state:
                state[k][:] = t[k].cpu().numpy().ravel()
        else:
            h = {k: v.reshape(1, -1) for k, v in arrs.items()}
            wa.driver_steps.update_swinterp_parameters(dims, **h)
    return f


def _hip_apply(wa, dims, dev):
    import torch

    def f(state, coszen_loc, albedo, out):
        src = {k: state[k] for k in od.APPLY_IN if k in state}
        src.update(coszen_loc=coszen_loc, albedo=albedo)
        arrs = {**src, **out}
        if dev:
            t = {k: torch.from_numpy(v.reshape(1, -1).copy()).cuda() for k, v in arrs.items()}
            wa.driver_steps.interp_sw_radiation(dims, **t)
            for k in out:
                out[k][:] = t[k].cpu().numpy().ravel()
        else:
            wa.driver_steps.interp_sw_radiation(dims, **{k: v.reshape(1, -1) for k, v in arrs.items()})
    return f


@pytest.mark.parametrize("dev", [True, False])
def test_swinterp_day_matches_reference(hip_lib, wa, dev):
    g = np.load(GOLD)
    dims = _dims(int(g["n"]), 1)
    # device pow / log differ from the reference's math library in the last bits
    check_day(_hip_update(wa, dims, dev), _hip_apply(wa, dims, dev), g, 2e-12, 1e-11)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_swinterp_window_and_real4(hip_lib, wa, dtype, tol):
    """A 7x5 tile inside a 11x9 memory window; points outside the tile stay untouched; REAL*4 storage follows the
    oracle's REAL*4 mode."""
    import torch
    rng = np.random.default_rng(11)
    ni, nj, halo = 7, 5, 2
    dims = _dims(ni, nj, halo)
    shp = (nj + 2 * halo, ni + 2 * halo)
    r4 = dtype == np.float32

    def field(lo, hi):
        return rng.uniform(lo, hi, shp).astype(dtype)

    inp = dict(coszen=field(0.0, 1.0), coszen_loc=field(0.0, 1.0), swddir=field(0, 600), swdown=field(0, 900))
    inp["coszen"][halo, halo] = 0.0
    state = dict(swddir_ref=field(0, 600), bb=field(-0.5, 2.5), Bx=field(-10, 900), swdown_ref=field(0, 900),
                 gg=field(-0.5, 2.5), Gx=field(-10, 900), coszen_ref=field(0.05, 1.0))
    keep = {k: v.copy() for k, v in state.items()}
    t = {k: torch.from_numpy(v.copy()).cuda() for k, v in {**inp, **state}.items()}
    wa.driver_steps.update_swinterp_parameters(dims, **t)
    got = {k: t[k].cpu().numpy() for k in state}
    sl = (slice(halo, halo + nj), slice(halo, halo + ni))
    o_in = {k: np.ascontiguousarray(v[sl].astype(np.float64).ravel()) for k, v in inp.items()}
    o_st = {k: np.ascontiguousarray(v[sl].astype(np.float64).ravel()) for k, v in state.items()}
    od.swinterp_update(o_in, o_st, real4=r4)
    for k in state:
        np.testing.assert_allclose(got[k][sl].ravel().astype(np.float64), o_st[k].astype(dtype).astype(np.float64),
                                   rtol=tol, atol=tol, err_msg=k)
        outside = np.ones(shp, bool); outside[sl] = False
        np.testing.assert_array_equal(got[k][outside], keep[k][outside], err_msg=k + " outside the tile")
    # interpolation on the updated state
    cl2, alb = field(0.0, 1.0), field(0.05, 0.8)
    outs = {k: field(1, 2) for k in od.APPLY_OUT}
    keep_o = {k: v.copy() for k, v in outs.items()}
    t2 = {k: t[k] for k in od.APPLY_IN if k in t}
    t2.update(coszen_loc=torch.from_numpy(cl2).cuda(), albedo=torch.from_numpy(alb).cuda())
    t2.update({k: torch.from_numpy(v.copy()).cuda() for k, v in outs.items()})
    wa.driver_steps.interp_sw_radiation(dims, **t2)
    st64 = {k: np.ascontiguousarray(got[k][sl].astype(np.float64).ravel()) for k in state}
    o_out = {k: np.zeros(ni * nj) for k in od.APPLY_OUT}
    od.swinterp_apply(st64, np.ascontiguousarray(cl2[sl].astype(np.float64).ravel()),
                      np.ascontiguousarray(alb[sl].astype(np.float64).ravel()), o_out, real4=r4)
    for k in od.APPLY_OUT:
        g2 = t2[k].cpu().numpy()
        np.testing.assert_allclose(g2[sl].ravel().astype(np.float64), o_out[k].astype(dtype).astype(np.float64),
                                   rtol=tol, atol=tol * 1e3 if r4 else tol, err_msg=k)
        outside = np.ones(shp, bool); outside[sl] = False
        np.testing.assert_array_equal(g2[outside], keep_o[k][outside], err_msg=k + " outside the tile")


@pytest.mark.parametrize("dev", [True, False])
def test_accumulate_buckets(hip_lib, wa, dev):
    import torch
    rng = np.random.default_rng(3)
    ni, nj = 9, 4
    dims = _dims(ni, nj)
    names = list(wa.driver_steps.LW_BUCKETS + wa.driver_steps.SW_BUCKETS)
    names.remove("swdntc")                               # an absent pair is skipped
    fl = {n: rng.uniform(0, 500, (nj, ni)) for n in names}
    ac = {n: rng.uniform(0, 1e7, (nj, ni)) for n in names}
    ref = {n: ac[n] + fl[n] * 18.0 for n in names}
    if dev:
        tf = {n: torch.from_numpy(v).cuda() for n, v in fl.items()}
        ta = {n: torch.from_numpy(v.copy()).cuda() for n, v in ac.items()}
        wa.driver_steps.accumulate_fluxes(dims, 18.0, tf, ta)
        got = {n: v.cpu().numpy() for n, v in ta.items()}
    else:
        got = {n: v.copy() for n, v in ac.items()}
        wa.driver_steps.accumulate_fluxes(dims, 18.0, fl, got)
    for n in names:
        np.testing.assert_allclose(got[n], ref[n], rtol=1e-15, atol=0, err_msg=n)   # the device may fuse a*b+c
