"""Child process of test_host_chunks_gpu.py::test_registered_host_arrays.  The parent sets RRTMG_HIP_REGISTER_HOST=1:
the caller's arrays are pinned in place and the chunks of a mem = 0 call travel by pitched copies straight between the
caller's tile and the device (csrc/tile_staging.h) -- no packing, no scatter.  Arrays that live for the whole process,
as a host model's do; every result must equal the packed path's bit for bit (the parent runs this script a second time
without the variable and compares the files), inside and outside the tile.  Prints REGISTER-OK."""
import importlib
import os
import sys

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [here, os.path.dirname(here)]


def main(tmp, out):
    wa = importlib.import_module("wrfchem-arc-interactions_amd")
    import bench
    ni, nj, nk = 24, 19, 40
    st = wa.synth.make_wrf_state(ni, nj, nk, seed=20261200, night_fraction=0.1)
    cloud = wa.synth_clouds.add_clouds(st, seed=20261201, cloudy_fraction=0.4)
    sw_tab, lw_tab = os.path.join(tmp, "RRTMG_SW_DATA"), os.path.join(tmp, "RRTMG_LW_DATA")
    wa.synth.write_sw_data(sw_tab, seed=20260001, real_bytes=8)
    wa.synth.write_lw_data(lw_tab, seed=20260003, real_bytes=8)
    wa.engine.rrtmg_swinit(sw_tab, device=0)
    wa.engine_lw.rrtmg_lwinit(5000.0, lw_tab, kme=nk + 1, device=0)
    keep = []   # every array handed to the engine stays alive until the process ends

    def arr(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a

    cld = {n: arr(st.a3[n]) for n in bench.CLD3}
    cld.update({n: arr(st.a2[n]) for n in bench.CLD2})
    flags = {k: v for k, v in cloud.items() if k != "absent"}
    sw_in = {n: arr(st.a2[n]) for n in bench.SW_IN2}
    sw_in.update({n: arr(st.a3[n]) for n in bench.ST_IN3 + bench.SW_AER})
    lw_in = {n: arr(st.a2[n]) for n in bench.LW_IN2}
    lw_in.update({n: sw_in[n] for n in bench.ST_IN3})
    lw_in.update({n: arr(st.a3[n]) for n in bench.LW_AER})
    res = {}
    for tag, tile in (("whole", {}), ("part", dict(its=3, ite=20, jts=2, jte=17))):   # a partial tile: pitched rows
        d = bench.dims_of(ni, nj, nk, tile)
        sw_out = {"rthratensw": arr(np.full((nj, nk + 1, ni), -7777.0))}
        sw_out.update({n: arr(np.full((nj, ni), -7777.0)) for n in bench.SW_OUT2})
        lw_out = {"rthratenlw": arr(np.full((nj, nk + 1, ni), -7777.0))}
        lw_out.update({n: arr(np.full((nj, ni), -7777.0)) for n in bench.LW_OUT2})
        for _ in range(2):   # the second call finds the arrays registered
            wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, **flags, **d, **sw_in, **cld, **sw_out)
            wa.engine_lw.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, **flags, **d, **lw_in, **cld, **lw_out)
        for n, v in list(sw_out.items()) + list(lw_out.items()):
            res[f"{tag}_{n}"] = v.copy()
        assert np.isfinite(sw_out["swupt"]).all() and (lw_out["lwupt"][sw_out["swupt"] != -7777.0] > 0).all()
    wa.engine_optics.optics_init(4, device=0)
    ms = wa.synth_mosaic.make_mosaic_state(st, nbin=4, seed=20261202)
    chem, alt, dz = arr(ms.chem), arr(ms.alt), arr(ms.dz8w)
    names = [f"{s}{w}" for s in ("tauaer", "waer", "gaer") for w in range(1, 5)] + [f"tauaerlw{b}" for b in range(1, 17)]
    oo = {n: arr(np.full(ms.alt.shape, -7777.0)) for n in names}
    d = bench.dims_of(ni, nj, nk, dict(its=2, ite=23, jts=1, jte=18, kts=1, kte=nk))
    wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, chem=chem, alt=alt, dz8w=dz, **oo)
    for n, v in oo.items():
        res["optics_" + n] = v.copy()
    # a host that frees an array tells the engine first (rrtmg_hip_unregister_host; a no-op for unknown addresses and
    # without the opt-in); the same call on the same arrays afterwards pins them again and gives the same bits
    wa.engine.unregister_host(*keep)
    wa.engine.unregister_host(np.zeros(3))
    wa.engine_optics.optical_averaging(idx=ms.idx, aer_op_opt=1, **d, chem=chem, alt=alt, dz8w=dz, **oo)
    for n, v in oo.items():
        np.testing.assert_array_equal(res["optics_" + n], v, err_msg="after unregister: " + n)
    np.savez(out, **res)
    print("REGISTER-OK")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
