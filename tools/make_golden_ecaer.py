#!/usr/bin/env python3
"""Golden vectors for aer_opt = 1 (iaer = 6, six ECMWF aerosol types): runs the reference rrtmg_sw
(oracle/_ref/ref_sw ... ecaer) on synthetic columns with synthetic type optical depths and stores inputs and
reference outputs in tests/golden/sw_cols_k40_ecaer.npz.  The clean-sky call is switched off: the reference leaves
its aerosol optical depth unset on this route.

Runs in the build container only (needs /root/reference); the fixture is committed."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrfchem_arc_amd as wa  # noqa: E402
from tools.make_golden import REF, GOLD, read_sw_out, sw_columns_from_state  # noqa: E402


def synth_ecaer(ncol, nlay, seed):
    """Optical depth at 0.55 um per layer and type: boundary-layer continental / maritime / urban, elevated desert
    dust, thin volcanic and stratospheric layers; a few aerosol-free columns."""
    rng = np.random.default_rng(seed)
    e = np.zeros((ncol, 6, nlay))
    z = np.arange(nlay) / nlay
    prof = [np.exp(-z / 0.08), np.exp(-z / 0.06), np.exp(-((z - 0.25) / 0.1) ** 2), np.exp(-z / 0.04),
            np.exp(-((z - 0.6) / 0.05) ** 2), np.exp(-((z - 0.75) / 0.08) ** 2)]
    for ic in range(ncol):
        if ic % 7 == 3:
            continue
        for ia in range(6):
            e[ic, ia] = rng.uniform(0, [0.2, 0.1, 0.3, 0.1, 0.02, 0.01][ia]) * prof[ia] / prof[ia].sum() * rng.integers(0, 2)
    e[:, :, -1] = 0.0   # the extra layer carries none (SW:11096)
    return e


def main():
    with tempfile.TemporaryDirectory() as td:
        tab = os.path.join(td, "RRTMG_SW_DATA")
        wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
        st = wa.synth.make_wrf_state(8, 4, 40, seed=20260013)
        cols = sw_columns_from_state(st, 0)
        ncol, nlay = cols["play"].shape
        ecaer = synth_ecaer(ncol, nlay, 20260014)
        fc, fo = os.path.join(td, "c.bin"), os.path.join(td, "o.bin")
        with open(fc, "wb") as f:
            f.write(np.array([ncol, nlay, 0, 1], dtype="<i4").tobytes())
            for ic in range(ncol):
                for k in ("play", "plev", "tlay", "tlev"):
                    f.write(cols[k][ic].astype("<f8").tobytes())
                f.write(np.float64(cols["tsfc"][ic]).tobytes())
                for k in ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr"):
                    f.write(cols[k][ic].astype("<f8").tobytes())
                f.write(cols["scal"][ic].astype("<f8").tobytes())
                for k in ("tauaer", "ssaaer", "asmaer"):      # read and ignored by the reference with iaer = 6
                    f.write(np.ascontiguousarray(cols[k][ic]).astype("<f8").tobytes())
                f.write(np.ascontiguousarray(ecaer[ic]).astype("<f8").tobytes())   # (type, layer) == Fortran (layer, type)
        subprocess.run([os.path.join(REF, "ref_sw"), tab, fc, fo, "-", "ecaer"], check=True, capture_output=True)
        ref = read_sw_out(fo, ncol, nlay)
    np.savez_compressed(os.path.join(GOLD, "sw_cols_k40_ecaer.npz"), table_seed=20260001,
                        **{"in_" + k: v for k, v in cols.items()}, in_ecaer=ecaer,
                        **{"ref_" + k: v for k, v in ref.items()})
    print("sw_cols_k40_ecaer", ecaer.shape, "swuflx TOA", ref["swuflx"][:3, -1])


if __name__ == "__main__":
    main()
