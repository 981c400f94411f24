"""Band-resolved aerosol input (tauaer3d_sw / ssaaer3d_sw / asyaer3d_sw of aer_opt = 2/3, SW:10982-10991): HIP
against the oracle, alone and together with the WRF-Chem arrays that override it where they qualify."""
import numpy as np
import pytest

from helpers import AER, IN2, IN3, alloc_outputs, compare_sw, dims_of

pytestmark = pytest.mark.gpu


def _aer4(st, seed):
    rng = np.random.default_rng(seed)
    shp = (14, st.nj, st.nk + 1, st.ni)
    tau = np.zeros(shp)
    tau[:, :, :st.nk, :] = rng.uniform(0.0, 0.03, size=(14, st.nj, st.nk, st.ni))
    tau[:, :, :, ::3] = 0.0                                  # some aerosol-free columns
    ssa = rng.uniform(0.7, 1.0, size=shp)
    asy = rng.uniform(0.5, 0.85, size=shp)
    return dict(tauaer3d_sw=tau, ssaaer3d_sw=ssa, asyaer3d_sw=asy)


@pytest.mark.parametrize("chem", [0, 1])
def test_band_resolved_aerosol_input(hip_lib, wa, sw_table_file, sw_oracle, chem):
    import torch
    from oracle import oracle as orc
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    st = wa.synth.make_wrf_state(6, 4, 40, seed=20260700 + chem)
    if chem:   # WRF-Chem arrays present too, but only in the lower half of the column
        for n in AER:
            if n.startswith("tauaer"):
                st.a3[n][:, 20:, :] = 0.0
    a4 = _aer4(st, 20260710)
    d = dims_of(st)
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + (AER if chem else [])})
    arrays.update(a4)
    ref = alloc_outputs(st)
    orc.swrad_tile(sw_oracle, d, arrays, 1370.0, chem, 1, ref)
    out = alloc_outputs(st)
    dev_in = {n: torch.from_numpy(np.ascontiguousarray(v)).cuda() for n, v in arrays.items()}
    dev_out = {n: torch.from_numpy(v).cuda() for n, v in out.items()}
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=chem, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
    wa.engine.sw_status()
    got = {n: v.cpu().numpy() for n, v in dev_out.items()}
    compare_sw(got, ref, 1e-9, 1e-9)
    # the input matters: aerosol columns differ from their clean-sky flux, aerosol-free columns do not
    has = a4["tauaer3d_sw"].sum(axis=(0, 2)) > 0
    assert np.abs(got["swuptcln"] - got["swupt"])[has].min() > 1e-3
    if not chem:
        np.testing.assert_array_equal(got["swuptcln"][~has], got["swupt"][~has])
    # host-pointer mode gives the same bits
    host_out = alloc_outputs(st)
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=chem, clean_atm_diag=1, icloud=0, **d,
                          **{n: np.ascontiguousarray(v) for n, v in arrays.items()}, **host_out)
    for n in got:
        np.testing.assert_array_equal(got[n], host_out[n], err_msg=n)
