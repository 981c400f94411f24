#!/usr/bin/env python3
"""Longwave golden fixtures: run the REFERENCE Fortran rrtmg_lw (oracle/_ref/ref_lw)
on synthetic RRTMG_LW_DATA tables and columns; see tools/make_golden.py."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrfchem_arc_amd as wa  # noqa: E402

synth = wa.synth
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
LW_COL_OUT = [("uflx", 1), ("dflx", 1), ("hr", 0), ("uflxc", 1), ("dflxc", 1), ("hrc", 0), ("uflxcln", 1), ("dflxcln", 1)]


def lw_columns_from_state(st, variant=0):
    """rrtmg_lw-level inputs with the buffer layers to 0 mb (an independent numpy construction,
    only to obtain plausible inputs)."""
    nk, ncol = st.nk, st.ncol
    nbuf = int(round(st.p_top * 0.01 / 4.0 + 1e-9))
    nlay = nk + nbuf

    def col3(name, n):
        return np.ascontiguousarray(np.transpose(st.a3[name][:, :n, :], (0, 2, 1)).reshape(ncol, n))

    plev = np.empty((ncol, nlay + 1))
    plev[:, :nk + 1] = col3("p8w", nk + 1) / 100.0
    for L in range(nk + 1, nlay + 1):
        plev[:, L] = plev[:, L - 1] - 4.0
    plev[:, nlay] = 0.0
    play = np.empty((ncol, nlay))
    play[:, :nk] = col3("p3d", nk) / 100.0
    play[:, nk:] = 0.5 * (plev[:, nk:-1] + plev[:, nk + 1:])
    tlev = np.empty((ncol, nlay + 1))
    tlev[:, :nk + 1] = col3("t8w", nk + 1)
    tlev[:, nk + 1:] = tlev[:, nk:nk + 1] + 1.2 * np.arange(1, nbuf + 1)[None, :]
    tlay = np.empty((ncol, nlay))
    tlay[:, :nk] = col3("t3d", nk)
    tlay[:, nk:] = 0.5 * (tlev[:, nk:-1] + tlev[:, nk + 1:])
    h2o = np.empty((ncol, nlay))
    h2o[:, :nk] = np.maximum(col3("qv3d", nk), 1e-12) * 1.607793
    h2o[:, nk:] = h2o[:, nk - 1:nk]
    o3 = 3e-8 + 8e-6 * np.exp(-((np.log(play) - np.log(8.0)) / 1.2) ** 2)
    ones = np.ones((ncol, nlay))
    cols = dict(play=play, plev=plev, tlay=tlay, tlev=tlev, tsfc=st.a2["tsk"].reshape(ncol), h2ovmr=h2o, o3vmr=o3,
                co2vmr=(379e-6 if not variant else 1500e-6) * ones, ch4vmr=1774e-9 * ones,
                n2ovmr=(319e-9 if not variant else 900e-9) * ones, o2vmr=0.209488 * ones,
                cfc11vmr=0.251e-9 * ones, cfc12vmr=0.538e-9 * ones, cfc22vmr=0.169e-9 * ones, ccl4vmr=0.093e-9 * ones)
    em = st.a2["emiss"].reshape(ncol)
    cols["emis"] = np.repeat(em[:, None], 16, axis=1) * (1.0 if not variant else np.linspace(0.9, 1.0, 16)[None, :])
    tau = np.zeros((ncol, 16, nlay))
    for b in range(16):
        tau[:, b, :nk] = col3(f"tauaerlw{b + 1}", nk)
    if variant:
        tau[::3] = 0.0
    cols["tauaer"] = tau
    return cols


def write_lw_columns(path, cols, clean=1, nrep=1):
    ncol, nlay = cols["play"].shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", ncol, nlay, clean, nrep))
        for ic in range(ncol):
            for k in ("play", "plev", "tlay", "tlev"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(struct.pack("<d", cols["tsfc"][ic]))
            for k in ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr", "cfc11vmr", "cfc12vmr", "cfc22vmr",
                      "ccl4vmr", "emis"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(np.ascontiguousarray(cols["tauaer"][ic]).astype("<f8").tobytes())


def read_lw_out(path, ncol, nlay):
    raw = np.fromfile(path, dtype="<f8")
    per = sum(nlay + e for _, e in LW_COL_OUT)
    raw = raw.reshape(ncol, per)
    out, o = {}, 0
    for name, e in LW_COL_OUT:
        n = nlay + e
        out[name] = raw[:, o:o + n].copy()
        o += n
    return out


def run_ref_lw(tables, cols, clean=1, nrep=1):
    exe = os.path.join(REF, "ref_lw")
    with tempfile.TemporaryDirectory() as td:
        fc, fo = os.path.join(td, "c.bin"), os.path.join(td, "o.bin")
        write_lw_columns(fc, cols, clean, nrep)
        r = subprocess.run([exe, tables, fc, fo], check=True, capture_output=True, text=True)
        return read_lw_out(fo, *cols["play"].shape), r.stdout


def main():
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        tab = os.path.join(td, "RRTMG_LW_DATA")
        synth.write_lw_data(tab, seed=20260003, real_bytes=8)
        for name, (ni, nj, nk, seed, variant) in {
            "lw_cols_k40": (6, 4, 40, 20260021, 0),
            "lw_cols_k50_variant": (3, 2, 50, 20260022, 1),
        }.items():
            st = synth.make_wrf_state(ni, nj, nk, seed=seed)
            cols = lw_columns_from_state(st, variant)
            ref, _ = run_ref_lw(tab, cols, clean=1)
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), table_seed=20260003,
                                **{"in_" + k: v for k, v in cols.items()},
                                **{"ref_" + k: v for k, v in ref.items()})
            print(name, cols["play"].shape, "OLR", ref["uflx"][:3, -1], "OLR clean", ref["uflxcln"][:3, -1])


if __name__ == "__main__":
    main()
