"""SSiB land surface (sf_surface_physics = 8, SW:10692-10703): spectral albedos where xland < 1.5, the broadband
albedo elsewhere.  HIP against the oracle (whose four-albedo column solver is pinned bitwise on the reference,
tests/golden/sw_cols_k50_variant.npz)."""
import numpy as np
import pytest

from helpers import AER, IN2, IN3, alloc_outputs, compare_sw, dims_of

pytestmark = pytest.mark.gpu


def test_ssib_spectral_albedos(hip_lib, wa, sw_table_file, sw_oracle):
    import torch
    from oracle import oracle as orc
    wa.engine.rrtmg_swinit(sw_table_file, device=0)
    st = wa.synth.make_wrf_state(7, 4, 40, seed=20260740)
    rng = np.random.default_rng(20260741)
    d = dims_of(st)
    arrays = {n: st.a2[n] for n in IN2}
    arrays.update({n: st.a3[n] for n in IN3 + AER})
    shp = st.a2["albedo"].shape
    arrays["xland"] = np.where(rng.uniform(size=shp) < 0.6, 1.0, 2.0)          # land / water
    arrays["alswvisdir"] = rng.uniform(0.03, 0.25, shp)
    arrays["alswvisdif"] = rng.uniform(0.03, 0.25, shp)
    arrays["alswnirdir"] = rng.uniform(0.15, 0.55, shp)
    arrays["alswnirdif"] = rng.uniform(0.15, 0.55, shp)
    ref = alloc_outputs(st)
    orc.swrad_tile(sw_oracle, d, arrays, 1370.0, 1, 1, ref)
    out = alloc_outputs(st)
    dev_in = {n: torch.from_numpy(np.ascontiguousarray(v)).cuda() for n, v in arrays.items()}
    dev_out = {n: torch.from_numpy(v).cuda() for n, v in out.items()}
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **dev_in, **dev_out)
    wa.engine.sw_status()
    got = {n: v.cpu().numpy() for n, v in dev_out.items()}
    compare_sw(got, ref, 1e-9, 1e-9)
    # water points keep the broadband solution, land points do not
    base = alloc_outputs(st)
    plain = {n: v for n, v in arrays.items() if not n.startswith("alsw")}
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d,
                          **{n: np.ascontiguousarray(v) for n, v in plain.items()}, **base)
    water = arrays["xland"] > 1.5
    np.testing.assert_array_equal(got["swupb"][water], base["swupb"][water])
    assert np.abs(got["swupb"] - base["swupb"])[~water].min() > 1e-3
    # host-pointer mode gives the same bits
    host_out = alloc_outputs(st)
    wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d,
                          **{n: np.ascontiguousarray(v) for n, v in arrays.items()}, **host_out)
    for n in got:
        np.testing.assert_array_equal(got[n], host_out[n], err_msg=n)
