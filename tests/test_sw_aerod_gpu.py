"""aer_opt = 1 (six ECMWF aerosol types, aerod(i,k,j,1:6), SW:9313-9340, 11083-11100): HIP against the oracle, whose
band mixture is pinned bitwise on the reference (tests/test_oracle_golden.py).  Every other aerosol input is
ignored on this route."""
import numpy as np
import pytest

from helpers import AER, IN2, IN3, alloc_outputs, compare_sw, dims_of

pytestmark = pytest.mark.gpu


def _aerod(st, seed):
    rng = np.random.default_rng(seed)
    e = np.zeros((6, st.nj, st.nk + 1, st.ni))
    z = np.arange(st.nk)[None, :, None] / st.nk
    for ia, (amp, scale) in enumerate([(0.2, 0.08), (0.1, 0.06), (0.3, 0.3), (0.1, 0.04), (0.02, 0.6), (0.01, 0.8)]):
        e[ia, :, :st.nk, :] = rng.uniform(0, amp, (st.nj, 1, st.ni)) * np.exp(-z / scale) / st.nk * 4
    e[:, :, :, ::4] = 0.0          # aerosol-free columns
    return e


@pytest.mark.parametrize("with_chem", [0, 1])
def test_ecmwf_aerosol_types(hip_lib, wa, sw_table_file, sw_oracle, with_chem):
    import torch
    from oracle import oracle as orc
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    st = wa.synth.make_wrf_state(6, 4, 40, seed=20260720 + with_chem)
    d = dims_of(st)
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + (AER if with_chem else [])})
    arrays["aerod"] = _aerod(st, 20260730)
    ref = alloc_outputs(st)
    orc.swrad_tile(sw_oracle, d, arrays, 1370.0, with_chem, 1, ref)
    out = alloc_outputs(st)
    dev_in = {n: torch.from_numpy(np.ascontiguousarray(v)).cuda() for n, v in arrays.items()}
    dev_out = {n: torch.from_numpy(v).cuda() for n, v in out.items()}
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=with_chem, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
    wa.engine.sw_status()
    got = {n: v.cpu().numpy() for n, v in dev_out.items()}
    compare_sw(got, ref, 1e-9, 1e-9)
    has = arrays["aerod"].sum(axis=(0, 2)) > 0
    assert np.abs(got["swuptcln"] - got["swupt"])[has].min() > 1e-4
    np.testing.assert_array_equal(got["swuptcln"][~has], got["swupt"][~has])
    if with_chem:   # the WRF-Chem arrays do not enter: same bits as without them
        arrays0 = {n: v for n, v in arrays.items() if n not in AER}
        out0 = alloc_outputs(st)
        wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=0, clean_atm_diag=1, icloud=0, **d,
                              **{n: np.ascontiguousarray(v) for n, v in arrays0.items()}, **out0)
        for n in got:
            np.testing.assert_array_equal(got[n], out0[n], err_msg=n)
