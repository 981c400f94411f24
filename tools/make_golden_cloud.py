#!/usr/bin/env python3
"""Golden vectors for the cloudy (McICA) shortwave path: runs the reference Fortran (oracle/_ref/ref_sw ... clouds:
mcica_subcol_sw + rrtmg_sw incl. cldprmc_sw, compiled from /root/reference by oracle/build_ref.sh) on synthetic
columns with synthetic in-cloud state and stores inputs + reference outputs in tests/golden/.

Runs in the build container only (needs /root/reference); the fixtures are committed."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrfchem_arc_amd as wa  # noqa: E402
from tools.make_golden import REF, GOLD, SW_COL_OUT, sw_columns_from_state  # noqa: E402

synth = wa.synth


def synth_cloud_columns(cols, seed, iceflag=3, cloudy_fraction=0.7):
    """In-cloud state per layer for every column of `cols` (the form RRTMG_SWRAD hands to mcica_subcol_sw):
    1-3 cloud decks of a few layers, liquid below ~ 600 mb, ice above, mixed in between; snow with iceflag 5."""
    rng = np.random.default_rng(seed)
    ncol, nlay = cols["play"].shape
    z = {k: np.zeros((ncol, nlay)) for k in ("cldfrac", "clwp", "ciwp", "cswp")}
    z["rel"] = np.full((ncol, nlay), 10.0)
    z["rei"] = np.full((ncol, nlay), 10.0)
    z["res"] = np.full((ncol, nlay), 10.0)
    for ic in range(ncol):
        if rng.uniform() > cloudy_fraction:
            continue
        for _ in range(rng.integers(1, 4)):
            k0 = int(rng.integers(1, nlay - 8))
            depth = int(rng.integers(1, 6))
            for k in range(k0, min(k0 + depth, nlay - 2)):
                p = cols["play"][ic, k]
                z["cldfrac"][ic, k] = rng.choice([1.0, rng.uniform(0.05, 0.95)])
                liq = np.clip((p - 350.0) / 300.0, 0.0, 1.0)
                z["clwp"][ic, k] = liq * rng.uniform(2.0, 80.0)
                z["ciwp"][ic, k] = (1.0 - liq) * rng.uniform(1.0, 40.0)
                if iceflag == 5:
                    z["cswp"][ic, k] = (1.0 - liq) * rng.uniform(0.0, 30.0) * (rng.uniform() < 0.6)
                z["rel"][ic, k] = rng.uniform(2.5, 20.0)
                z["rei"][ic, k] = rng.uniform(6.0, 135.0)
                z["res"][ic, k] = rng.uniform(20.0, 130.0)
    z["cldflags"] = np.tile(np.array([5 if iceflag == 5 else 2, iceflag, 1], dtype=np.int32), (ncol, 1))
    return z


def write_cloud_columns(path, cols, clean=1, nrep=1):
    ncol, nlay = cols["play"].shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", ncol, nlay, clean, nrep))
        for ic in range(ncol):
            for k in ("play", "plev", "tlay", "tlev"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(struct.pack("<d", cols["tsfc"][ic]))
            for k in ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(cols["scal"][ic].astype("<f8").tobytes())
            for k in ("tauaer", "ssaaer", "asmaer"):
                f.write(np.ascontiguousarray(cols[k][ic]).astype("<f8").tobytes())
            for k in ("cldfrac", "clwp", "ciwp", "cswp", "rel", "rei", "res"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(cols["cldflags"][ic].astype("<i4").tobytes())


def run_ref_sw_cloud(tables, cols, clean=1, nrep=1):
    exe = os.path.join(REF, "ref_sw")
    ncol, nlay = cols["play"].shape
    with tempfile.TemporaryDirectory() as td:
        fc, fo = os.path.join(td, "c.bin"), os.path.join(td, "o.bin")
        write_cloud_columns(fc, cols, clean, nrep)
        r = subprocess.run([exe, tables, fc, fo, "-", "clouds"], check=True, capture_output=True, text=True)
        raw = np.fromfile(fo, dtype="<f8")
        per = sum(nlay + e for _, e in SW_COL_OUT) + 112 * nlay
        raw = raw.reshape(ncol, per)
        out, o = {}, 0
        for name, e in SW_COL_OUT:
            n = nlay + e
            out[name] = raw[:, o:o + n].copy()
            o += n
        # Fortran cldfmcl(ngptsw, nlay) written g fastest -> [layer][g]; store as [g][layer]
        out["cldfmc"] = raw[:, o:].reshape(ncol, nlay, 112).transpose(0, 2, 1).copy()
        return out, r.stdout


def run_ref_lw_cloud(tables, cols, clean=1):
    from tools.make_golden_lw import LW_COL_OUT
    exe = os.path.join(REF, "ref_lw")
    ncol, nlay = cols["play"].shape
    with tempfile.TemporaryDirectory() as td:
        fc, fo = os.path.join(td, "c.bin"), os.path.join(td, "o.bin")
        with open(fc, "wb") as f:
            f.write(struct.pack("<4i", ncol, nlay, clean, 1))
            for ic in range(ncol):
                for k in ("play", "plev", "tlay", "tlev"):
                    f.write(cols[k][ic].astype("<f8").tobytes())
                f.write(struct.pack("<d", cols["tsfc"][ic]))
                for k in ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr",
                          "cfc22vmr", "ccl4vmr", "emis"):
                    f.write(cols[k][ic].astype("<f8").tobytes())
                f.write(np.ascontiguousarray(cols["tauaer"][ic]).astype("<f8").tobytes())
                for k in ("cldfrac", "clwp", "ciwp", "cswp", "rel", "rei", "res"):
                    f.write(cols[k][ic].astype("<f8").tobytes())
                f.write(cols["cldflags"][ic].astype("<i4").tobytes())
        subprocess.run([exe, tables, fc, fo, "clouds"], check=True, capture_output=True, text=True)
        raw = np.fromfile(fo, dtype="<f8")
        per = sum(nlay + e for _, e in LW_COL_OUT) + 140 * nlay
        raw = raw.reshape(ncol, per)
        out, o = {}, 0
        for name, e in LW_COL_OUT:
            n = nlay + e
            out[name] = raw[:, o:o + n].copy()
            o += n
        out["cldfmc"] = raw[:, o:].reshape(ncol, nlay, 140).transpose(0, 2, 1).copy()
        return out


def main_lw():
    from tools.make_golden_lw import lw_columns_from_state
    with tempfile.TemporaryDirectory() as td:
        tab = os.path.join(td, "RRTMG_LW_DATA")
        synth.write_lw_data(tab, seed=20260003, real_bytes=8)
        for name, (ni, nj, nk, seed, iceflag) in {
            "lw_cloud_cols_k40_ice3": (6, 4, 40, 20260031, 3),
            "lw_cloud_cols_k50_ice5": (4, 3, 50, 20260032, 5),
        }.items():
            st = synth.make_wrf_state(ni, nj, nk, seed=seed)
            cols = lw_columns_from_state(st, 0)
            cl = synth_cloud_columns(cols, seed + 100, iceflag)
            # no cloud in the buffer layers above the model top (LW:12543-12553); liquid radius >= 2.5 um (LW cldprmc)
            for k in ("cldfrac", "clwp", "ciwp", "cswp"):
                cl[k][:, nk:] = 0.0
            cols.update(cl)
            ref = run_ref_lw_cloud(tab, cols, clean=1)
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), table_seed=20260003,
                                **{"in_" + k: v for k, v in cols.items()},
                                **{"ref_" + k: v for k, v in ref.items()})
            cloudy = (ref["cldfmc"].sum(axis=(1, 2)) > 0).sum()
            print(name, "columns", cols["play"].shape, "cloudy columns", cloudy,
                  "max |total-clear| OLR", np.abs(ref["uflx"][:, -1] - ref["uflxc"][:, -1]).max())


def main():
    os.makedirs(GOLD, exist_ok=True)
    main_lw()
    with tempfile.TemporaryDirectory() as td:
        tab = os.path.join(td, "RRTMG_SW_DATA")
        synth.write_sw_data(tab, seed=20260001, real_bytes=8)
        for name, (ni, nj, nk, seed, iceflag) in {
            "sw_cloud_cols_k40_ice3": (6, 4, 40, 20260021, 3),
            "sw_cloud_cols_k50_ice5": (4, 3, 50, 20260022, 5),
        }.items():
            st = synth.make_wrf_state(ni, nj, nk, seed=seed)
            cols = sw_columns_from_state(st, 0)
            cols.update(synth_cloud_columns(cols, seed + 100, iceflag))
            ref, _ = run_ref_sw_cloud(tab, cols, clean=1)
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), table_seed=20260001,
                                **{"in_" + k: v for k, v in cols.items()},
                                **{"ref_" + k: v for k, v in ref.items()})
            cloudy = (ref["cldfmc"].sum(axis=(1, 2)) > 0).sum()
            print(name, "columns", cols["play"].shape, "cloudy columns", cloudy,
                  "max |total-clear| TOA up", np.abs(ref["swuflx"][:, -1] - ref["swuflxc"][:, -1]).max())


if __name__ == "__main__":
    main()
